// ORACLE -- TEST INFRASTRUCTURE ONLY (see orb_oracle.hpp header).  PARITY UNPINNED.
//
// Matcher half of the oracle: restates src/ORBmatcher.cc, the grid part of src/FrameKTL.cc and the
// all-pairs knn-2 semantics of include/utils.h:81-111 (cv::BFMatcher(NORM_HAMMING)::knnMatch, k=2).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "orb_oracle.hpp"

namespace orc {

// ORBmatcher::DescriptorDistance: src/ORBmatcher.cc:1794-1810 (bit-hack popcount over 8 int32 words)
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    unsigned int v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// knnMatch(k=2) of include/utils.h:100-101: per query the two smallest distances over the allowed
// train rows; OpenCV's batchDistance keeps the lower train index on ties (strict `<` insertion) [OCV-RECALL].
// idx = -1 / d = -1 where fewer than 1 / 2 train rows are allowed.
void knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* mask, int32_t* idx0, int32_t* d0, int32_t* idx1, int32_t* d1) {
  for (int i = 0; i < nq; ++i) {
    int bi0 = -1, bd0 = 1 << 30, bi1 = -1, bd1 = 1 << 30;
    for (int j = 0; j < nt; ++j) {
      if (mask && !mask[(size_t)i * nt + j]) continue;
      int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < bd0) {
        bd1 = bd0, bi1 = bi0;
        bd0 = d, bi0 = j;
      } else if (d < bd1) {
        bd1 = d, bi1 = j;
      }
    }
    idx0[i] = bi0, d0[i] = bi0 < 0 ? -1 : bd0;
    idx1[i] = bi1, d1[i] = bi1 < 0 ? -1 : bd1;
  }
}

// MapPoint::ComputeDistinctiveDescriptors: src/MapPoint.cc:197-270 (distance matrix :236-247, least-median pick :250-263)
int distinctive_descriptor(const uint8_t* desc, int N, int* best_median) {
  if (N <= 0) {
    *best_median = -1;
    return -1;
  }
  std::vector<std::vector<float>> Distances(N, std::vector<float>(N));
  for (int i = 0; i < N; i++) {
    Distances[i][i] = 0;
    for (int j = i + 1; j < N; j++) {
      int distij = descriptor_distance(desc + (size_t)i * 32, desc + (size_t)j * 32);
      Distances[i][j] = (float)distij;
      Distances[j][i] = (float)distij;
    }
  }
  int BestMedian = 0x7fffffff, BestIdx = 0;
  for (int i = 0; i < N; i++) {
    std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
    std::sort(vDists.begin(), vDists.end());
    int median = vDists[(size_t)(0.5 * (N - 1))];
    if (median < BestMedian) {
      BestMedian = median;
      BestIdx = i;
    }
  }
  *best_median = BestMedian;
  return BestIdx;
}

// grid constants: include/FrameKTL.h:45-46
static const int GRID_ROWS = 48, GRID_COLS = 64;

// FrameKTL ctor :83-84 (inverse cell size), compute_descriptors :250-264 (fill), PosInGrid :426-436 (uses round())
void FrameGrid::build(const KeyPoint* _kps, int _n, int _minX, int _minY, int _maxX, int _maxY) {
  kps = _kps, n = _n, minX = _minX, minY = _minY, maxX = _maxX, maxY = _maxY;
  invW = (float)GRID_COLS / (float)(maxX - minX);
  invH = (float)GRID_ROWS / (float)(maxY - minY);
  cells.assign((size_t)GRID_COLS * GRID_ROWS, {});
  for (int i = 0; i < n; ++i) {
    int posX = (int)roundf((kps[i].x - minX) * invW);
    int posY = (int)roundf((kps[i].y - minY) * invH);
    if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
    cells[(size_t)posX * GRID_ROWS + posY].push_back(i);
  }
}

// FrameKTL::GetFeaturesInArea: src/FrameKTL.cc:359-424
std::vector<int> FrameGrid::GetFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel) const {
  std::vector<int> vIndices;
  int nMinCellX = (int)floorf((x - minX - r) * invW);
  nMinCellX = std::max(0, nMinCellX);
  if (nMinCellX >= GRID_COLS) return vIndices;
  int nMaxCellX = (int)ceilf((x - minX + r) * invW);
  nMaxCellX = std::min(GRID_COLS - 1, nMaxCellX);
  if (nMaxCellX < 0) return vIndices;
  int nMinCellY = (int)floorf((y - minY - r) * invH);
  nMinCellY = std::max(0, nMinCellY);
  if (nMinCellY >= GRID_ROWS) return vIndices;
  int nMaxCellY = (int)ceilf((y - minY + r) * invH);
  nMaxCellY = std::min(GRID_ROWS - 1, nMaxCellY);
  if (nMaxCellY < 0) return vIndices;
  bool bCheckLevels = true, bSameLevel = false;
  if (minLevel == -1 && maxLevel == -1)
    bCheckLevels = false;
  else if (minLevel == maxLevel)
    bSameLevel = true;
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const std::vector<int>& vCell = cells[(size_t)ix * GRID_ROWS + iy];
      for (size_t j = 0; j < vCell.size(); j++) {
        const KeyPoint& kpUn = kps[vCell[j]];
        if (bCheckLevels && !bSameLevel) {
          if (kpUn.octave < minLevel || kpUn.octave > maxLevel) continue;
        } else if (bSameLevel) {
          if (kpUn.octave != minLevel) continue;
        }
        if (fabsf(kpUn.x - x) > r || fabsf(kpUn.y - y) > r) continue;
        vIndices.push_back(vCell[j]);
      }
    }
  return vIndices;
}

// ORBmatcher::SearchByProjection(FrameKTL&, const vector<MapPoint*>&, th): src/ORBmatcher.cc:49-125
// RadiusByViewingCos :127-133.  TH_HIGH = 100 (:40).
int search_by_projection(const FrameGrid& g, const uint8_t* fdesc, int32_t* assigned, int nmp, const float* projx, const float* projy,
                         const int32_t* level, const float* viewcos, const uint8_t* inview, const uint8_t* mpdesc,
                         const float* scaleFactors, float th, float nnratio) {
  const int TH_HIGH = 100;
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < nmp; iMP++) {
    if (!inview[iMP]) continue;  // mbTrackInView / isBad
    const int nPredictedLevel = level[iMP];
    float r = ((double)viewcos[iMP] > 0.998) ? 2.5f : 4.0f;
    if (bFactor) r *= th;
    std::vector<int> vNear = g.GetFeaturesInArea(projx[iMP], projy[iMP], r * scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    if (vNear.empty()) continue;
    const uint8_t* MPdescriptor = mpdesc + (size_t)iMP * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : vNear) {
      if (assigned[idx] >= 0) continue;
      const int dist = descriptor_distance(MPdescriptor, fdesc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestLevel2 = bestLevel;
        bestLevel = g.kps[idx].octave;
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = g.kps[idx].octave;
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      assigned[bestIdx] = iMP;
      nmatches++;
    }
  }
  return nmatches;
}

}  // namespace orc
