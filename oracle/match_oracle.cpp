// ORACLE -- TEST INFRASTRUCTURE ONLY (see orb_oracle.hpp header).  PARITY UNPINNED.
//
// Matcher half of the oracle: restates src/ORBmatcher.cc, the grid part of src/FrameKTL.cc and the
// all-pairs knn-2 semantics of include/utils.h:81-111 (cv::BFMatcher(NORM_HAMMING)::knnMatch, k=2).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

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


static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // src/ORBmatcher.cc:40-42

// ORBmatcher::ComputeThreeMaxima: src/ORBmatcher.cc:1748-1789
void compute_three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = sizes[i];
    if (s > max1) {
      max3 = max2;
      max2 = max1;
      max1 = s;
      ind3 = ind2;
      ind2 = ind1;
      ind1 = i;
    } else if (s > max2) {
      max3 = max2;
      max2 = s;
      ind3 = ind2;
      ind2 = i;
    } else if (s > max3) {
      max3 = s;
      ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    ind2 = -1;
    ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
}

// rotation-bin idiom shared by the search loops, e.g. :1709-1716
static int rot_bin(float a1, float a2) {
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}
// the tail every loop ends with, e.g. :1724-1743: entries of all but the three fullest bins are removed
template <class Remove>
static int apply_rot_hist(std::vector<int>* rotHist, Remove remove) {
  int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH], removed = 0;
  for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
  compute_three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
  for (int i = 0; i < HISTO_LENGTH; i++) {
    if (i == ind1 || i == ind2 || i == ind3) continue;
    for (size_t j = 0; j < rotHist[i].size(); j++) {
      remove(rotHist[i][j]);
      removed++;
    }
  }
  return removed;
}

// ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): src/ORBmatcher.cc:1672-1745
int search_by_projection_kf(const FrameGrid& g, const uint8_t* fdesc, int32_t* assigned, int nmp, const float* u, const float* v,
                            const int32_t* level, const uint8_t* valid, const uint8_t* mpdesc, const float* kf_angle,
                            const float* scaleFactors, float th, int ORBdist, bool checkOri) {
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < nmp; i++) {
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    float radius = th * scaleFactors[nPredictedLevel];
    std::vector<int> vIndices2 = g.GetFeaturesInArea(u[i], v[i], radius, nPredictedLevel - 1, nPredictedLevel + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = mpdesc + (size_t)i * 32;
    int bestDist = 0x7fffffff, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (assigned[i2] >= 0) continue;
      int dist = descriptor_distance(dMP, fdesc + (size_t)i2 * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx2 = i2;
      }
    }
    if (bestDist <= ORBdist) {
      assigned[bestIdx2] = i;
      nmatches++;
      if (checkOri) rotHist[rot_bin(kf_angle[i], g.kps[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (checkOri) nmatches -= apply_rot_hist(rotHist, [&](int idx2) { assigned[idx2] = -1; });
  return nmatches;
}

// merge walk of :178-249 / :741-819 / :876-965 (std::map iteration; lower_bound jumps == sorted merge)
// ---- the four ORBmatcher members without a caller in the reference, as literal loops ----
static void mat_Rp_plus_t(const float* R, const float* p, const float* t, float* out);

// ORBmatcher::WindowSearch(F1, F2, windowSize, vpMapPointMatches2, minScaleLevel, maxScaleLevel): src/ORBmatcher.cc:409-516.
// has_mp1[i1] = F1.mvpMapPoints[i1] non-null and not bad.  match21[i2] = vnMatches21 (index into F1, or -1) -- vpMapPointMatches2[i2]
// is F1's map point at that index.
int window_search(const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* has_mp1, const FrameGrid& g2, const uint8_t* desc2, int n2,
                  int windowSize, int minScaleLevel, int maxScaleLevel, float nnratio, bool checkOri, int32_t* match21) {
  int nmatches = 0;
  for (int i = 0; i < n2; i++) match21[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const bool bMinLevel = minScaleLevel > 0;
  const bool bMaxLevel = maxScaleLevel < 0x7fffffff;
  for (int i1 = 0; i1 < n1; i1++) {
    if (!has_mp1[i1]) continue;
    const KeyPoint& k1 = kp1[i1];
    int level1 = k1.octave;
    if (bMinLevel)
      if (level1 < minScaleLevel) continue;
    if (bMaxLevel)
      if (level1 > maxScaleLevel) continue;
    std::vector<int> vIndices2 = g2.GetFeaturesInArea(k1.x, k1.y, windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const uint8_t* d1 = desc1 + (size_t)i1 * 32;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (match21[i2] >= 0) continue;  // vpMapPointMatches2[i2]
      int dist = descriptor_distance(d1, desc2 + (size_t)i2 * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestIdx2 = i2;
      } else if (dist < bestDist2) {
        bestDist2 = dist;
      }
    }
    if (bestDist <= bestDist2 * nnratio && bestDist <= TH_HIGH) {
      match21[bestIdx2] = i1;
      nmatches++;
      rotHist[rot_bin(k1.angle, g2.kps[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (checkOri) nmatches -= apply_rot_hist(rotHist, [&](int idx2) { match21[idx2] = -1; });
  return nmatches;
}

// ORBmatcher::SearchByProjection(F1, F2, windowSize, vpMapPointMatches2): :519-594.  usable1[i1] = map point of F1's keypoint i1
// exists, is not bad and is not among F2's map points already; assigned2[i2] in/out: >= 0 where vpMapPointMatches2[i2] is set
// (on return: the F1 index for new matches).  Camera = F2's pose and intrinsics.
int search_by_projection_frames(const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* usable1, const float* xyz1, const Camera& F2,
                                const FrameGrid& g2, const uint8_t* desc2, int32_t* assigned2, int windowSize, float nnratio) {
  int nmatches = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    if (!usable1[i1]) continue;
    int level1 = kp1[i1].octave;
    float x3Dc2[3];
    mat_Rp_plus_t(F2.Rcw, xyz1 + 3 * (size_t)i1, F2.tcw, x3Dc2);
    const float xc2 = x3Dc2[0], yc2 = x3Dc2[1];
    const float invzc2 = 1.0 / x3Dc2[2];
    float u2 = F2.fx * xc2 * invzc2 + F2.cx;
    float v2 = F2.fy * yc2 * invzc2 + F2.cy;
    std::vector<int> vIndices2 = g2.GetFeaturesInArea(u2, v2, windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const uint8_t* d1 = desc1 + (size_t)i1 * 32;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (assigned2[i2] >= 0) continue;
      int dist = descriptor_distance(d1, desc2 + (size_t)i2 * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestIdx2 = i2;
      } else if (dist < bestDist2) {
        bestDist2 = dist;
      }
    }
    if (static_cast<float>(bestDist) <= static_cast<float>(bestDist2) * nnratio && bestDist <= TH_HIGH) {
      assigned2[bestIdx2] = i1;
      nmatches++;
    }
  }
  return nmatches;
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): :598-713.  prev_matched: 2 floats per F1
// keypoint, updated at the end (:705-708).
int search_for_initialization(const KeyPoint* kp1, int n1, const uint8_t* desc1, const FrameGrid& g2, const uint8_t* desc2, int n2,
                              float* prev_matched, int32_t* vnMatches12, int windowSize, float nnratio, bool checkOri) {
  int nmatches = 0;
  for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vMatchedDistance(n2, 0x7fffffff);
  std::vector<int> vnMatches21(n2, -1);
  for (int i1 = 0; i1 < n1; i1++) {
    const KeyPoint k1 = kp1[i1];
    int level1 = k1.octave;
    if (level1 > 0) continue;
    std::vector<int> vIndices2 = g2.GetFeaturesInArea(prev_matched[2 * i1], prev_matched[2 * i1 + 1], windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const uint8_t* d1 = desc1 + (size_t)i1 * 32;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      int dist = descriptor_distance(d1, desc2 + (size_t)i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestIdx2 = i2;
      } else if (dist < bestDist2) {
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) {
          vnMatches12[vnMatches21[bestIdx2]] = -1;
          nmatches--;
        }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (checkOri) rotHist[rot_bin(k1.angle, g2.kps[bestIdx2].angle)].push_back(i1);
      }
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
    compute_three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        int idx1 = rotHist[i][j];
        if (vnMatches12[idx1] >= 0) {
          vnMatches12[idx1] = -1;
          nmatches--;
        }
      }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (vnMatches12[i1] >= 0) prev_matched[2 * i1] = g2.kps[vnMatches12[i1]].x, prev_matched[2 * i1 + 1] = g2.kps[vnMatches12[i1]].y;
  return nmatches;
}

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th): :1507-1620.  usable_last[i] = LastFrame.mvpMapPoints[i] non-null and
// !LastFrame.mvbOutlier[i]; octave_last = LastFrame.mvKeys[i].octave, angle_last = LastFrame.mvKeysUn[i].angle.  assigned in/out:
// >= 0 where CurrentFrame.mvpMapPoints[i2] is set (new matches hold the LastFrame index).
int search_by_projection_last(const Camera& Cur, const FrameGrid& g, const uint8_t* fdesc, int32_t* assigned, int nlast, const uint8_t* usable_last,
                              const float* xyz_last, const int32_t* octave_last, const float* angle_last, const uint8_t* desc_last,
                              const float* scaleFactors, float th, bool checkOri) {
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < nlast; i++) {
    if (!usable_last[i]) continue;
    float x3Dc[3];
    mat_Rp_plus_t(Cur.Rcw, xyz_last + 3 * (size_t)i, Cur.tcw, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = 1.0 / x3Dc[2];
    float u = Cur.fx * xc * invzc + Cur.cx;
    float v = Cur.fy * yc * invzc + Cur.cy;
    if (u < Cur.minX || u > Cur.maxX) continue;
    if (v < Cur.minY || v > Cur.maxY) continue;
    int nPredictedOctave = octave_last[i];
    float radius = th * scaleFactors[nPredictedOctave];
    std::vector<int> vIndices2 = g.GetFeaturesInArea(u, v, radius, nPredictedOctave - 1, nPredictedOctave + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = desc_last + (size_t)i * 32;
    int bestDist = 0x7fffffff, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (assigned[i2] >= 0) continue;
      int dist = descriptor_distance(dMP, fdesc + (size_t)i2 * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx2 = i2;
      }
    }
    if (bestDist <= TH_HIGH) {
      assigned[bestIdx2] = i;
      nmatches++;
      if (checkOri) rotHist[rot_bin(angle_last[i], g.kps[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (checkOri) nmatches -= apply_rot_hist(rotHist, [&](int idx2) { assigned[idx2] = -1; });
  return nmatches;
}

template <class F>
static void walk_shared_nodes(const FeatureVector& a, const FeatureVector& b, F f) {
  int i = 0, j = 0;
  while (i < a.n_nodes && j < b.n_nodes) {
    if (a.node[i] == b.node[j]) {
      f(i, j);
      i++, j++;
    } else if (a.node[i] < b.node[j]) {
      while (i < a.n_nodes && a.node[i] < b.node[j]) i++;  // lower_bound
    } else {
      while (j < b.n_nodes && b.node[j] < a.node[i]) j++;
    }
  }
}

int search_by_bow(bool kf_kf, const FeatureVector& fv1, int n1, const uint8_t* desc1, const float* angle1, const uint8_t* usable1,
                  const FeatureVector& fv2, int n2, const uint8_t* desc2, const float* angle2, const uint8_t* usable2, float nnratio,
                  bool checkOri, int32_t* match12) {
  for (int i = 0; i < n1; i++) match12[i] = -1;
  std::vector<char> taken2(n2, 0);  // vpMapPointMatches[realIdxF] != NULL (:199)  /  vbMatched2 (:766)
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  walk_shared_nodes(fv1, fv2, [&](int a, int b) {
    for (int e1 = fv1.start[a]; e1 < fv1.start[a + 1]; e1++) {
      const int idx1 = fv1.feat[e1];
      if (!usable1[idx1]) continue;  // :188-194 / :751-757
      const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
      int bestDist1 = 0x7fffffff, bestIdx2 = -1, bestDist2 = 0x7fffffff;
      for (int e2 = fv2.start[b]; e2 < fv2.start[b + 1]; e2++) {
        const int idx2 = fv2.feat[e2];
        if (taken2[idx2]) continue;
        if (kf_kf && usable2 && !usable2[idx2]) continue;  // :768-772
        const int dist = descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
        if (dist < bestDist1) {
          bestDist2 = bestDist1;
          bestDist1 = dist;
          bestIdx2 = idx2;
        } else if (dist < bestDist2) {
          bestDist2 = dist;
        }
      }
      const bool under = kf_kf ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW;  // :786 vs :216
      if (under && (float)bestDist1 < nnratio * (float)bestDist2) {
        match12[idx1] = bestIdx2;
        taken2[bestIdx2] = 1;
        if (checkOri) rotHist[rot_bin(angle1[idx1], angle2[bestIdx2])].push_back(idx1);
        nmatches++;
      }
    }
  });
  if (checkOri) nmatches -= apply_rot_hist(rotHist, [&](int idx1) { match12[idx1] = -1; });
  return nmatches;
}

// ORBmatcher::CheckDistEpipolarLine: src/ORBmatcher.cc:136-153
static bool check_dist_epipolar_line(const KeyPoint& kp1, const KeyPoint& kp2, const float* F12, const float* sigma2) {
  const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
  const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
  const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
  const float num = a * kp2.x + b * kp2.y + c;
  const float den = a * a + b * b;
  if (den == 0) return false;
  const float dsqr = num * num / den;
  return dsqr < 3.84 * sigma2[kp2.octave];
}

int search_for_triangulation(const FeatureVector& fv1, const KeyPoint* kp1, int n1, const uint8_t* desc1, const uint8_t* has_mp1,
                             const FeatureVector& fv2, const KeyPoint* kp2, int n2, const uint8_t* desc2, const uint8_t* has_mp2,
                             const float* F12, const float* sigma2, bool checkOri, int32_t* match12) {
  int nmatches = 0;
  std::vector<char> vbMatched2(n2, 0);
  for (int i = 0; i < n1; i++) match12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  walk_shared_nodes(fv1, fv2, [&](int a, int b) {
    for (int e1 = fv1.start[a]; e1 < fv1.start[a + 1]; e1++) {
      const int idx1 = fv1.feat[e1];
      if (has_mp1[idx1]) continue;
      const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
      std::vector<std::pair<int, size_t>> vDistIndex;
      for (int e2 = fv2.start[b]; e2 < fv2.start[b + 1]; e2++) {
        const int idx2 = fv2.feat[e2];
        if (vbMatched2[idx2] || has_mp2[idx2]) continue;
        const int dist = descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
        if (dist > TH_LOW) continue;
        vDistIndex.push_back(std::make_pair(dist, (size_t)idx2));
      }
      if (vDistIndex.empty()) continue;
      std::sort(vDistIndex.begin(), vDistIndex.end());
      int BestDist = vDistIndex.front().first;
      int DistTh = (int)round(2 * BestDist);
      for (size_t id = 0; id < vDistIndex.size(); id++) {
        if (vDistIndex[id].first > DistTh) break;
        int currentIdx2 = (int)vDistIndex[id].second;
        if (check_dist_epipolar_line(kp1[idx1], kp2[currentIdx2], F12, sigma2)) {
          vbMatched2[currentIdx2] = 1;
          match12[idx1] = currentIdx2;
          nmatches++;
          if (checkOri) rotHist[rot_bin(kp1[idx1].angle, kp2[currentIdx2].angle)].push_back(idx1);
          break;
        }
      }
    }
  });
  if (checkOri) nmatches -= apply_rot_hist(rotHist, [&](int idx1) { match12[idx1] = -1; });
  return nmatches;
}

// Fuse(pKF, vpMapPoints, th): src/ORBmatcher.cc:1077-1101; KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:952-992) has no level
// filter of its own (expressed here as -1,-1), the level test is :1094
void fuse_search(const FrameGrid& g, const uint8_t* kfdesc, int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid,
                 const uint8_t* mpdesc, const float* scaleFactors, float th, int32_t* best_idx, int32_t* best_dist) {
  for (int i = 0; i < nmp; i++) {
    best_idx[i] = -1, best_dist[i] = -1;
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    const float radius = th * scaleFactors[nPredictedLevel];
    std::vector<int> vIndices = g.GetFeaturesInArea(u[i], v[i], radius, -1, -1);
    if (vIndices.empty()) continue;
    int bestDist = 0x7fffffff, bestIdx = -1;
    for (int idx : vIndices) {
      const int kpLevel = g.kps[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = descriptor_distance(mpdesc + (size_t)i * 32, kfdesc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = idx;
      }
    }
    if (bestDist <= TH_LOW) best_idx[i] = bestIdx, best_dist[i] = bestDist;
  }
}

// ---- cv::Mat arithmetic used by the projection code, under the assumptions stated in orb_oracle.hpp ----
static void mat_Rp_plus_t(const float* R, const float* p, const float* t, float* out) {  // cv::gemm small-matrix path
  for (int i = 0; i < 3; i++) {
    const float t0 = R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2];
    out[i] = (float)(t0 * 1.0 + t[i] * 1.0);
  }
}
static float mat_norm3(const float* a) {  // cv::norm(NORM_L2), CV_32F: double accumulator
  double s = 0;
  for (int i = 0; i < 3; i++) {
    double v = a[i];
    s += v * v;
  }
  return (float)std::sqrt(s);
}
static double mat_dot3(const float* a, const float* b) {  // cv::Mat::dot, CV_32F: double accumulator
  double r = 0;
  for (int i = 0; i < 3; i++) r += (double)a[i] * b[i];
  return r;
}

bool is_in_frustum(const Camera& F, const float* P, const float* Pn, float mfMinDistance, float mfMaxDistance, float viewingCosLimit,
                   float scaleFactor, int nScaleLevels, float* u_out, float* v_out, int* level, float* viewCos_out) {
  float Pc[3];
  mat_Rp_plus_t(F.Rcw, P, F.tcw, Pc);
  const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
  if (PcZ < 0.0) return false;
  const float invz = 1.0 / PcZ;
  const float u = F.fx * PcX * invz + F.cx;
  const float v = F.fy * PcY * invz + F.cy;
  if (u < F.minX || u > F.maxX) return false;
  if (v < F.minY || v > F.maxY) return false;
  const float maxDistance = 1.2f * mfMaxDistance;  // GetMaxDistanceInvariance src/MapPoint.cc:350-354
  const float minDistance = 0.8f * mfMinDistance;
  const float PO[3] = {P[0] - F.Ow[0], P[1] - F.Ow[1], P[2] - F.Ow[2]};
  const float dist = mat_norm3(PO);
  if (dist < minDistance || dist > maxDistance) return false;
  float viewCos = mat_dot3(PO, Pn) / dist;
  if (viewCos < viewingCosLimit) return false;
  // PredictScale: ratio = mfMaxDistance/currentDist; nScale = ceil(log(ratio)/mfLogScaleFactor) -- float overloads
  const float ratio = mfMaxDistance / dist;
  const float mfLogScaleFactor = logf(scaleFactor);
  int nScale = (int)ceilf(logf(ratio) / mfLogScaleFactor);
  if (nScale < 0)
    nScale = 0;
  else if (nScale >= nScaleLevels)
    nScale = nScaleLevels - 1;
  *u_out = u, *v_out = v, *level = nScale, *viewCos_out = viewCos;
  return true;
}

bool project_kf_reloc(const Camera& F, const float* x3Dw, float mfMinDistance, const float* scaleFactors, int nScaleLevels, float* u_out,
                      float* v_out, int* level) {
  float Ow[3];  // Ow = -Rcw.t()*tcw :1628
  for (int c = 0; c < 3; c++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)F.Rcw[3 * k + c] * (double)F.tcw[k];
    Ow[c] = (float)(s * -1.0);
  }
  float x3Dc[3];
  mat_Rp_plus_t(F.Rcw, x3Dw, F.tcw, x3Dc);
  const float xc = x3Dc[0], yc = x3Dc[1];
  const float invzc = 1.0 / x3Dc[2];
  float u = F.fx * xc * invzc + F.cx;
  float v = F.fy * yc * invzc + F.cy;
  if (u < F.minX || u > F.maxX) return false;
  if (v < F.minY || v > F.maxY) return false;
  float minDistance = 0.8f * mfMinDistance;
  const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
  float dist3D = mat_norm3(PO);
  float ratio = dist3D / minDistance;
  const float* it = std::lower_bound(scaleFactors, scaleFactors + nScaleLevels, ratio);
  *level = std::min((int)(it - scaleFactors), nScaleLevels - 1);
  *u_out = u, *v_out = v;
  return true;
}

bool project_fuse(const Camera& K, const float* p3Dw, const float* Pn, float mfMinDistance, float mfMaxDistance, const float* scaleFactors,
                  int nScaleLevels, float* u_out, float* v_out, int* level) {
  const int nMaxLevel = nScaleLevels - 1;
  float p3Dc[3];
  mat_Rp_plus_t(K.Rcw, p3Dw, K.tcw, p3Dc);
  if (p3Dc[2] < 0.0f) return false;
  const float invz = 1 / p3Dc[2];
  const float x = p3Dc[0] * invz;
  const float y = p3Dc[1] * invz;
  const float u = K.fx * x + K.cx;
  const float v = K.fy * y + K.cy;
  if (!(u >= K.minX && u < K.maxX && v >= K.minY && v < K.maxY)) return false;  // KeyFrame::IsInImage src/KeyFrame.cc:994-997
  const float maxDistance = 1.2f * mfMaxDistance;
  const float minDistance = 0.8f * mfMinDistance;
  const float PO[3] = {p3Dw[0] - K.Ow[0], p3Dw[1] - K.Ow[1], p3Dw[2] - K.Ow[2]};
  const float dist3D = mat_norm3(PO);
  if (dist3D < minDistance || dist3D > maxDistance) return false;
  if (mat_dot3(PO, Pn) < 0.5 * dist3D) return false;
  const float ratio = dist3D / minDistance;
  const float* it = std::lower_bound(scaleFactors, scaleFactors + nScaleLevels, ratio);
  *level = std::min((int)(it - scaleFactors), nMaxLevel);
  *u_out = u, *v_out = v;
  return true;
}

// ---- Sim3 forms (loop closing) ----
// Decomposition of Scw at the head of SearchByProjection(pKF, Scw, ...) and Fuse(pKF, Scw, ...): src/ORBmatcher.cc:299-303 / :1145-1149.
// cv::Mat expressions, as OpenCV 3.4 evaluates them (unpinned, orb_oracle.hpp): Mat::dot accumulates in double; `M / s` is
// convertTo(alpha = 1./s) whose 32F kernel multiplies by (float)alpha; `-A.t() * b` is gemm(GEMM_1_T, alpha = -1) on the general
// path (double accumulation).  Scw: 3 rows of row_stride floats (a 4x4 or 3x4 row-major matrix).
void sim3_decompose(const float* Scw, int row_stride, float* Rcw, float* tcw, float* Ow) {
  double dot = 0;
  for (int k = 0; k < 3; k++) dot += (double)Scw[k] * Scw[k];
  const float scw = sqrt(dot);
  const float inv = (float)(1. / scw);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) Rcw[3 * i + j] = Scw[i * row_stride + j] * inv;
    tcw[i] = Scw[i * row_stride + 3] * inv;
  }
  for (int c = 0; c < 3; c++) {
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += (double)Rcw[3 * k + c] * (double)tcw[k];
    Ow[c] = (float)(acc * -1.0);
  }
}

// SearchBySim3 :1284-1287: sR12 = s12*R12 (convertTo, float product); sR21 = (1.0/s12)*R12.t() (transpose, then convertTo with
// (float)(1.0/s12)); t21 = -sR21*t12 (gemm small-matrix path: fp32 row sum, times alpha = -1 in double)
void sim3_relative(float s12, const float* R12, const float* t12, float* sR12, float* sR21, float* t21) {
  for (int i = 0; i < 9; i++) sR12[i] = R12[i] * s12;
  const float inv = (float)(1.0 / s12);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) sR21[3 * i + j] = R12[3 * j + i] * inv;
  for (int i = 0; i < 3; i++) {
    const float t0 = sR21[3 * i] * t12[0] + sR21[3 * i + 1] * t12[1] + sR21[3 * i + 2] * t12[2];
    t21[i] = (float)(t0 * -1.0 + 0.f * 0.0);
  }
}

// SearchBySim3, the per-point prologue of either direction (:1323-1359 / :1403-1441): world -> own camera (Ra, ta) -> other
// camera (sR, t) -> pixel of the other key frame K (intrinsics + image bounds are the only members of K that are read)
bool project_sim3(const float* Ra, const float* ta, const float* sR, const float* t, const Camera& K, const float* p3Dw, float minDistance,
                  float maxDistance, const float* scaleFactors, int nScaleLevels, float* u_out, float* v_out, int* level) {
  float p3Dc1[3], p3Dc2[3];
  mat_Rp_plus_t(Ra, p3Dw, ta, p3Dc1);
  mat_Rp_plus_t(sR, p3Dc1, t, p3Dc2);
  if (p3Dc2[2] < 0.0) return false;
  float invz = 1.0 / p3Dc2[2];
  float x = p3Dc2[0] * invz;
  float y = p3Dc2[1] * invz;
  float u = K.fx * x + K.cx;
  float v = K.fy * y + K.cy;
  if (!(u >= K.minX && u < K.maxX && v >= K.minY && v < K.maxY)) return false;  // KeyFrame::IsInImage src/KeyFrame.cc:994-997
  float dist3D = mat_norm3(p3Dc2);
  if (dist3D < minDistance || dist3D > maxDistance) return false;
  float ratio = dist3D / minDistance;
  const float* it = std::lower_bound(scaleFactors, scaleFactors + nScaleLevels, ratio);
  *level = std::min((int)(it - scaleFactors), nScaleLevels - 1);
  *u_out = u, *v_out = v;
  return true;
}

// SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) :357-398 after the projection tests: matched[idx] >= 0 stands for
// vpMatched[idx] != NULL; a new match stores the candidate's index
int search_by_projection_sim3(const FrameGrid& g, const uint8_t* kfdesc, int32_t* matched, int nmp, const float* u, const float* v,
                              const int32_t* level, const uint8_t* valid, const uint8_t* mpdesc, const float* scaleFactors, int th) {
  int nmatches = 0;
  for (int iMP = 0; iMP < nmp; iMP++) {
    if (!valid[iMP]) continue;
    const int nPredictedLevel = level[iMP];
    const float radius = th * scaleFactors[nPredictedLevel];
    std::vector<int> vIndices = g.GetFeaturesInArea(u[iMP], v[iMP], radius, -1, -1);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = mpdesc + (size_t)iMP * 32;
    int bestDist = 0x7fffffff, bestIdx = -1;
    for (int idx : vIndices) {
      if (matched[idx] >= 0) continue;
      const int kpLevel = g.kps[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = descriptor_distance(dMP, kfdesc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = idx;
      }
    }
    if (bestDist <= TH_LOW) {
      matched[bestIdx] = iMP;
      nmatches++;
    }
  }
  return nmatches;
}

// one direction of SearchBySim3 (:1361-1394 / :1443-1476): best key point of the other frame on levels [l-1, l], <= TH_HIGH
static void sim3_direction(const FrameGrid& g, const uint8_t* kfdesc, int n, const float* u, const float* v, const int32_t* level,
                           const uint8_t* valid, const uint8_t* mpdesc, const float* scaleFactors, float th, std::vector<int>& vnMatch) {
  const int TH_HIGH = 100;  // :40
  vnMatch.assign(n, -1);
  for (int i = 0; i < n; i++) {
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    float radius = th * scaleFactors[nPredictedLevel];
    std::vector<int> vIndices = g.GetFeaturesInArea(u[i], v[i], radius, -1, -1);
    if (vIndices.empty()) continue;
    int bestDist = 0x7fffffff, bestIdx = -1;
    for (int idx : vIndices) {
      const int octave = g.kps[idx].octave;
      if (octave < nPredictedLevel - 1 || octave > nPredictedLevel) continue;
      int dist = descriptor_distance(mpdesc + (size_t)i * 32, kfdesc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx = idx;
      }
    }
    if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
  }
}

// SearchBySim3 :1361-1504.  valid12[i1] = point i1 of KF1 exists, is not already matched, is not bad and passed the projection into
// KF2 (u12, v12, level12); likewise valid21 for KF2's points projected into KF1.  match12[i1] = index in KF2 or -1 (the reference
// stores vpMapPoints2[that index]).
int search_by_sim3(const FrameGrid& g1, const uint8_t* desc1, int n1, const FrameGrid& g2, const uint8_t* desc2, int n2, const float* u12,
                   const float* v12, const int32_t* level12, const uint8_t* valid12, const uint8_t* mpdesc1, const float* u21, const float* v21,
                   const int32_t* level21, const uint8_t* valid21, const uint8_t* mpdesc2, const float* scaleFactors1,
                   const float* scaleFactors2, float th, int32_t* match12) {
  std::vector<int> vnMatch1, vnMatch2;
  sim3_direction(g2, desc2, n1, u12, v12, level12, valid12, mpdesc1, scaleFactors2, th, vnMatch1);
  sim3_direction(g1, desc1, n2, u21, v21, level21, valid21, mpdesc2, scaleFactors1, th, vnMatch2);
  int nFound = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    match12[i1] = -1;
    int idx2 = vnMatch1[i1];
    if (idx2 >= 0) {
      int idx1 = vnMatch2[idx2];
      if (idx1 == i1) {
        match12[i1] = idx2;
        nFound++;
      }
    }
  }
  return nFound;
}

// ---- DBoW2 ----
void bow_transform_one(const Vocabulary& voc, const uint8_t* feature, int levelsup, int* word_id, double* weight, int* nid_out) {
  const int nid_level = voc.L - levelsup;
  int nid = -1;
  if (nid_level <= 0) nid = 0;  // root
  int final_id = 0;             // root
  int current_level = 0;
  while (voc.child_start[final_id] != voc.child_start[final_id + 1]) {  // do { ... } while(!isLeaf) on a tree whose root has children
    ++current_level;
    const int32_t* nodes = voc.children + voc.child_start[final_id];
    const int nn = voc.child_start[final_id + 1] - voc.child_start[final_id];
    final_id = nodes[0];
    double best_d = descriptor_distance(feature, voc.descriptor + (size_t)final_id * 32);  // FORB::distance == DescriptorDistance
    for (int c = 1; c < nn; ++c) {
      const int id = nodes[c];
      double d = descriptor_distance(feature, voc.descriptor + (size_t)id * 32);
      if (d < best_d) {
        best_d = d;
        final_id = id;
      }
    }
    if (current_level == nid_level) nid = final_id;
  }
  if (nid < 0) nid = final_id;
  *word_id = voc.word_id[final_id];
  *weight = voc.weight[final_id];
  *nid_out = nid;
}

void bow_transform(const Vocabulary& voc, const uint8_t* features, int n, int levelsup, std::vector<std::pair<uint32_t, double>>& bow_out,
                   std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& fv_out) {
  std::map<uint32_t, double> v;
  std::map<uint32_t, std::vector<uint32_t>> fv;
  const bool must = voc.normalize != 0;
  if (voc.weighting == 0 || voc.weighting == 1) {
    for (int i = 0; i < n; ++i) {
      int id, nid;
      double w;
      bow_transform_one(voc, features + (size_t)i * 32, levelsup, &id, &w, &nid);
      if (w > 0) {
        auto it = v.lower_bound((uint32_t)id);  // BowVector::addWeight
        if (it != v.end() && !(v.key_comp()((uint32_t)id, it->first)))
          it->second += w;
        else
          v.insert(it, std::make_pair((uint32_t)id, w));
        fv[(uint32_t)nid].push_back((uint32_t)i);
      }
    }
    if (!v.empty() && !must) {
      const double nd = v.size();
      for (auto& kv : v) kv.second /= nd;
    }
  } else {
    for (int i = 0; i < n; ++i) {
      int id, nid;
      double w;
      bow_transform_one(voc, features + (size_t)i * 32, levelsup, &id, &w, &nid);
      if (w > 0) {
        auto it = v.lower_bound((uint32_t)id);  // addIfNotExist
        if (it == v.end() || v.key_comp()((uint32_t)id, it->first)) v.insert(it, std::make_pair((uint32_t)id, w));
        fv[(uint32_t)nid].push_back((uint32_t)i);
      }
    }
  }
  if (must) {
    double norm = 0.0;
    if (voc.normalize == 1) {
      for (auto& kv : v) norm += fabs(kv.second);
    } else {
      for (auto& kv : v) norm += kv.second * kv.second;
      norm = sqrt(norm);
    }
    if (norm > 0.0)
      for (auto& kv : v) kv.second /= norm;
  }
  bow_out.assign(v.begin(), v.end());
  fv_out.assign(fv.begin(), fv.end());
}

void haloc_hash(const float* r_, int num_proj, int r_stride, const uint8_t* desc, int rows, float* hash) {
  for (int k = 0; k < num_proj * 32; ++k) hash[k] = 0.0f;
  if (rows == 0) return;
  unsigned k = 0;
  for (int i = 0; i < num_proj; i++) {
    for (int n = 0; n < 32; n++) {
      float desc_sum = 0.0;
      for (int m = 0; m < rows; m++) desc_sum += r_[(size_t)i * r_stride + m] * (float)desc[(size_t)m * 32 + n];
      hash[k] = desc_sum / (float)rows;
      k++;
    }
  }
}

}  // namespace orc
