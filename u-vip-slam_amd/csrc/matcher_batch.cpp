// Batched forms of the two matcher loops of the LocalMapping thread.
//
//   CreateNewMapPoints (src/LocalMapping.cc:1058-1180) calls SearchForTriangulation(current KF, neighbour k) for up to 20 neighbours;
//   between two calls it triangulates pair k's matches and gives the accepted ones map points, which makes those features of the
//   current key frame ineligible for pair k + 1 (`if(pMP1) continue;`, src/ORBmatcher.cc:885-889).  The descriptor distances and the
//   epipolar test do not depend on that: uvo_search_for_triangulation_batch computes them for every pair in ONE launch and one host
//   wait; uvo_search_for_triangulation_next(k, has_mp1 now) then replays the reference's acceptance loop (:886-960) for pair k on the
//   host, in the reference's order -- the same result as 20 single calls, with one device round trip instead of 20.
//
//   SearchInNeighbors (src/LocalMapping.cc:1228-1236) calls Fuse(target k, the current key frame's map points) per target.  Fuse has no
//   exclusivity among the map points, so uvo_fuse_batch computes the projection tests and the best key point of every (target, map
//   point) in one pass (one upload of the map points, per target: grid + projection + walk, all stream-ordered, one download, one
//   wait); the map mutation between targets stays with the caller (include/uvo/compat/ORBmatcher.h: FuseTargets).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "matcher_priv.hpp"

namespace uvo {
void launch_group_dist_pairs(hipStream_t s, int nq, int total, const int32_t* d_cand_start, const int32_t* d_cand_idx, const uint8_t* d_qdesc,
                             const uint8_t* d_tdesc, const int32_t* d_tlevel, const int32_t* d_q_pair, const int32_t* d_pair_base, const float* d_f12,
                             const float* d_qx, const float* d_qy, const float* d_tx, const float* d_ty, const float* d_sigma2, int sig_stride,
                             uint32_t* d_cand);
void launch_fuse_walk(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y, int nmp,
                      const uint8_t* d_valid, const float* d_u, const float* d_v, const int32_t* d_level, const uint8_t* d_mp_desc, const float* d_sf, float th,
                      int32_t* d_cell_start, int32_t* d_cell_items, int32_t* d_cell_of_kp, int32_t* d_best_idx, int32_t* d_best_dist);
void launch_project(hipStream_t s, int mode, const uvo_camera_pose& cam, int n, const float* d_xyz, const float* d_normal, const float* d_min,
                    const float* d_max, const float* d_max_raw, const uint8_t* d_usable, const float* d_sf, int nlevels, float log_sf, float cos_limit,
                    uint8_t* d_valid, float* d_u, float* d_v, int32_t* d_level, float* d_cos);

// what uvo_search_for_triangulation_batch leaves in the handle for the _next calls
struct TriBatch {
  int n1 = 0, n_pairs = 0;
  std::vector<uint8_t> has_mp1;   // as given to _batch: a feature that had a map point then is never a query
  std::vector<float> angle1;
  struct Pair {
    int n2 = 0, q_begin = 0, q_end = 0;  // this pair's queries [q_begin, q_end) in the reference's visiting order
    std::vector<float> angle2;
  };
  std::vector<Pair> pairs;
  std::vector<int32_t> q_idx1, start;  // query -> feature of key frame 1; CSR offsets into cand
  std::vector<uint32_t> cand;          // packed: idx2 | distance << 16 | octave << 25 | epipolar ok << 31
};
void tri_batch_free(void* p) { delete static_cast<TriBatch*>(p); }
}  // namespace uvo

using namespace uvo;

namespace {
enum BatchSlot { B0 = 24, B_QDESC = B0, B_TDESC, B_TLEVEL, B_START, B_CIDX, B_QPAIR, B_MISC, B_CAND };  // scratch slots of the batched forms

int ensure(uvo_matcher* m, int slot, size_t bytes, void** out) {
  DevBuf& b = m->scratch[slot];
  if (bytes > b.cap) {
    if (b.p) {
      if (hipStreamSynchronize(m->stream) != hipSuccess) return matcher_fail(UVO_E_HIP, "hipStreamSynchronize failed");
      (void)hipFree(b.p);
      b.p = nullptr, b.cap = 0;
    }
    const size_t want = bytes + bytes / 2 + 256;
    uint8_t* p = nullptr;
    int rc = m_alloc(&p, want);
    if (rc) return rc;
    b.p = p, b.cap = want;
  }
  *out = b.p;
  return UVO_OK;
}
template <class T>
int upload(uvo_matcher* m, int slot, const T* src, size_t count, T** dev) {
  void* p = nullptr;
  int rc = ensure(m, slot, std::max<size_t>(count, 1) * sizeof(T), &p);
  if (rc) return rc;
  *dev = static_cast<T*>(p);
  if (count && src && hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, m->stream) != hipSuccess) return matcher_fail(UVO_E_HIP, "hipMemcpyAsync failed");
  return UVO_OK;
}
#define RC(call)                   \
  do {                             \
    const int _rc = (call);        \
    if (_rc != UVO_OK) return _rc; \
  } while (0)

int check_fv(const uvo_feature_vector* fv, int n) {
  if (!fv || fv->n_nodes < 0) return matcher_fail(UVO_E_BADARG, "null feature vector");
  if (fv->n_nodes == 0) return UVO_OK;
  if (!fv->node || !fv->start || !fv->feat) return matcher_fail(UVO_E_BADARG, "null feature vector arrays");
  for (int k = 0; k < fv->n_nodes; ++k) {
    if (k && fv->node[k] <= fv->node[k - 1]) return matcher_fail(UVO_E_BADARG, "feature vector node ids must be strictly ascending");
    if (fv->start[k + 1] < fv->start[k]) return matcher_fail(UVO_E_BADARG, "feature vector offsets must be non-decreasing");
  }
  for (int e = fv->start[0]; e < fv->start[fv->n_nodes]; ++e)
    if (fv->feat[e] < 0 || fv->feat[e] >= n) return matcher_fail(UVO_E_BADARG, "feature index outside the keypoint range");
  return UVO_OK;
}

// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1748-1789) on bin populations
void three_maxima(const int* hist, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = hist[i];
    if (s > max1) {
      max3 = max2, max2 = max1, max1 = s;
      ind3 = ind2, ind2 = ind1, ind1 = i;
    } else if (s > max2) {
      max3 = max2, max2 = s;
      ind3 = ind2, ind2 = i;
    } else if (s > max3) {
      max3 = s, ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    ind2 = -1, ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
}
}  // namespace

extern "C" {

int uvo_search_for_triangulation_batch(uvo_matcher* m, const uvo_feature_vector* fv1, const uvo_keypoint* kp1, int n1, const uint8_t* desc1,
                                       const uint8_t* has_mp1, int n_pairs, const uvo_triangulation_pair* pairs) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  tri_batch_free(m->tri_batch);
  m->tri_batch = nullptr;
  if (n1 < 0 || n_pairs < 0 || n_pairs > 4096) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (n_pairs > 0 && !pairs) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (n1 > 0 && (!kp1 || !desc1 || !has_mp1)) return matcher_fail(UVO_E_BADARG, "null pointer");
  RC(check_fv(fv1, n1));
  TriBatch* tb = new TriBatch();
  struct Guard {
    TriBatch* t;
    ~Guard() { delete t; }
  } guard{tb};
  tb->n1 = n1, tb->n_pairs = n_pairs;
  tb->has_mp1.assign(has_mp1, has_mp1 + n1);
  tb->angle1.resize(n1);
  for (int i = 0; i < n1; ++i) tb->angle1[i] = kp1[i].angle;
  tb->pairs.resize(n_pairs);
  tb->start.assign(1, 0);
  std::vector<int32_t> cidx, q_pair, pair_base(std::max(n_pairs, 1), 0), tlevel;
  std::vector<float> tx, ty, f12((size_t)std::max(n_pairs, 1) * 9, 0.f);
  int sig_stride = 1;
  for (int p = 0; p < n_pairs; ++p) sig_stride = std::max(sig_stride, pairs[p].nlevels);
  std::vector<float> sigma((size_t)std::max(n_pairs, 1) * sig_stride, 0.f);
  size_t nt = 0;
  for (int p = 0; p < n_pairs; ++p) {
    const uvo_triangulation_pair& P = pairs[p];
    if (P.n2 < 0 || P.n2 > 65535 || P.nlevels < 1 || P.nlevels > 64) return matcher_fail(UVO_E_BADARG, "bad pair (at most 65535 keypoints, 1 <= nlevels <= 64)");
    if (P.n2 > 0 && (!P.kp2 || !P.desc2 || !P.has_mp2 || !P.sigma2)) return matcher_fail(UVO_E_BADARG, "null pointer in a pair");
    RC(check_fv(P.fv2, P.n2));
    for (int k = 0; k < P.n2; ++k)
      if (P.kp2[k].octave < 0 || P.kp2[k].octave >= P.nlevels) return matcher_fail(UVO_E_BADARG, "keypoint level outside the pair's sigma table");
    TriBatch::Pair& Q = tb->pairs[p];
    Q.n2 = P.n2, Q.q_begin = (int)tb->q_idx1.size();
    Q.angle2.resize(P.n2);
    pair_base[p] = (int32_t)nt;
    for (int k = 0; k < P.n2; ++k) {
      Q.angle2[k] = P.kp2[k].angle;
      tx.push_back(P.kp2[k].x), ty.push_back(P.kp2[k].y), tlevel.push_back(P.kp2[k].octave);
    }
    memcpy(&f12[(size_t)p * 9], P.f12, 9 * sizeof(float));
    memcpy(&sigma[(size_t)p * sig_stride], P.sigma2, (size_t)P.nlevels * sizeof(float));
    // queries in the reference's visiting order: shared nodes ascending, features of key frame 1 in node order (:886-892); candidates =
    // the node's features of key frame 2 without a map point (`|| pMP2`, :903-905), in node order
    int a = 0, b = 0;
    while (n1 > 0 && P.n2 > 0 && a < fv1->n_nodes && b < P.fv2->n_nodes) {
      if (fv1->node[a] == P.fv2->node[b]) {
        for (int e = fv1->start[a]; e < fv1->start[a + 1]; ++e) {
          const int idx1 = fv1->feat[e];
          if (has_mp1[idx1]) continue;
          tb->q_idx1.push_back(idx1);
          q_pair.push_back(p);
          for (int e2 = P.fv2->start[b]; e2 < P.fv2->start[b + 1]; ++e2) {
            const int idx2 = P.fv2->feat[e2];
            if (!P.has_mp2[idx2]) cidx.push_back((int32_t)nt + idx2);
          }
          tb->start.push_back((int32_t)cidx.size());
        }
        ++a, ++b;
      } else if (fv1->node[a] < P.fv2->node[b]) {
        ++a;
      } else {
        ++b;
      }
    }
    Q.q_end = (int)tb->q_idx1.size();
    nt += (size_t)P.n2;
  }
  const int nq = (int)tb->q_idx1.size(), total = (int)cidx.size();
  tb->cand.assign((size_t)total, 0u);
  if (total > 0) {
    UVO_HIP_CHECK(hipSetDevice(m->device));
    // the uploads below read local staging vectors asynchronously: whatever way this block is left, the stream is drained first
    struct Drain {
      hipStream_t s;
      bool armed;
      ~Drain() {
        if (armed) (void)hipStreamSynchronize(s);
      }
    } drain{m->stream, true};
    std::vector<uint8_t> qdesc((size_t)nq * 32), tdesc(nt * 32);
    std::vector<float> qx(nq), qy(nq);
    for (int i = 0; i < nq; ++i) {
      memcpy(&qdesc[(size_t)i * 32], desc1 + (size_t)tb->q_idx1[i] * 32, 32);
      qx[i] = kp1[tb->q_idx1[i]].x, qy[i] = kp1[tb->q_idx1[i]].y;
    }
    for (int p = 0; p < n_pairs; ++p)
      if (pairs[p].n2 > 0) memcpy(&tdesc[(size_t)pair_base[p] * 32], pairs[p].desc2, (size_t)pairs[p].n2 * 32);
    // one float block: qx | qy | tx | ty | f12 | sigma
    std::vector<float> fl;
    fl.insert(fl.end(), qx.begin(), qx.end());
    fl.insert(fl.end(), qy.begin(), qy.end());
    fl.insert(fl.end(), tx.begin(), tx.end());
    fl.insert(fl.end(), ty.begin(), ty.end());
    fl.insert(fl.end(), f12.begin(), f12.end());
    fl.insert(fl.end(), sigma.begin(), sigma.end());
    std::vector<int32_t> il(q_pair);
    il.insert(il.end(), pair_base.begin(), pair_base.end());
    uint8_t *d_qdesc, *d_tdesc;
    int32_t *d_tlevel, *d_start, *d_cidx, *d_il;
    float* d_fl;
    uint32_t* d_cand;
    RC(upload(m, B_QDESC, qdesc.data(), qdesc.size(), &d_qdesc));
    RC(upload(m, B_TDESC, tdesc.data(), tdesc.size(), &d_tdesc));
    RC(upload(m, B_TLEVEL, tlevel.data(), tlevel.size(), &d_tlevel));
    RC(upload(m, B_START, tb->start.data(), tb->start.size(), &d_start));
    RC(upload(m, B_CIDX, cidx.data(), cidx.size(), &d_cidx));
    RC(upload(m, B_QPAIR, il.data(), il.size(), &d_il));
    RC(upload(m, B_MISC, fl.data(), fl.size(), &d_fl));
    RC(upload<uint32_t>(m, B_CAND, nullptr, (size_t)total, &d_cand));
    const float *d_qx = d_fl, *d_qy = d_qx + nq, *d_tx = d_qy + nq, *d_ty = d_tx + nt, *d_f12 = d_ty + nt, *d_sigma = d_f12 + f12.size();
    launch_group_dist_pairs(m->stream, nq, total, d_start, d_cidx, d_qdesc, d_tdesc, d_tlevel, d_il, d_il + nq, d_f12, d_qx, d_qy, d_tx, d_ty, d_sigma, sig_stride,
                            d_cand);
    UVO_HIP_CHECK(hipGetLastError());
    UVO_HIP_CHECK(hipMemcpyAsync(tb->cand.data(), d_cand, (size_t)total * 4, hipMemcpyDeviceToHost, m->stream));
    UVO_HIP_CHECK(hipStreamSynchronize(m->stream));  // the only host wait of the batch
    drain.armed = false;
  }
  guard.t = nullptr;
  m->tri_batch = tb;
  return UVO_OK;
}

int uvo_search_for_triangulation_next(uvo_matcher* m, int pair, const uint8_t* has_mp1_now, int check_orientation, int32_t* match12, int* n_matches) {
  if (!m || !n_matches) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_matches = 0;
  const TriBatch* tb = static_cast<const TriBatch*>(m->tri_batch);
  if (!tb) return matcher_fail(UVO_E_BADARG, "no batch: call uvo_search_for_triangulation_batch first");
  if (pair < 0 || pair >= tb->n_pairs) return matcher_fail(UVO_E_BADARG, "pair outside the batch");
  if (tb->n1 == 0) return UVO_OK;
  if (!match12 || !has_mp1_now) return matcher_fail(UVO_E_BADARG, "null pointer");
  for (int i = 0; i < tb->n1; ++i) {
    match12[i] = -1;
    // the batch only prepared the features that had no map point when it began; one that LOST its point since would be a query now
    if (tb->has_mp1[i] && !has_mp1_now[i]) return matcher_fail(UVO_E_BADARG, "a feature lost its map point since the batch began: start a new batch");
  }
  const TriBatch::Pair& P = tb->pairs[pair];
  std::vector<uint8_t> vbMatched2(P.n2, 0);
  std::vector<std::pair<int, uint32_t> > vDistIndex;  // (distance, idx2 | ok << 31): sorts like the reference's pair<int, size_t>
  int nmatches = 0;
  constexpr int HISTO_LENGTH = 30, TH_LOW = 50;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  for (int q = P.q_begin; q < P.q_end; ++q) {
    const int idx1 = tb->q_idx1[q];
    if (has_mp1_now[idx1]) continue;  // :885-889
    vDistIndex.clear();
    for (int e = tb->start[q]; e < tb->start[q + 1]; ++e) {
      const uint32_t w = tb->cand[e];
      const uint32_t idx2 = w & 0xffffu;
      const int dist = (int)((w >> 16) & 0x1ffu);
      if (vbMatched2[idx2]) continue;  // :903-905 (the map-point half of the test was applied when the lists were built)
      if (dist > TH_LOW) continue;     // :911-912
      vDistIndex.push_back(std::make_pair(dist, idx2 | (w & 0x80000000u)));
    }
    if (vDistIndex.empty()) continue;
    // sort(vDistIndex): by distance, then by idx2 -- the flag in bit 31 must not take part
    std::sort(vDistIndex.begin(), vDistIndex.end(), [](const std::pair<int, uint32_t>& a, const std::pair<int, uint32_t>& b) {
      return a.first != b.first ? a.first < b.first : (a.second & 0x7fffffffu) < (b.second & 0x7fffffffu);
    });
    const int BestDist = vDistIndex.front().first;
    const int DistTh = (int)std::round((double)(2 * BestDist));  // :921
    for (size_t id = 0; id < vDistIndex.size(); id++) {
      if (vDistIndex[id].first > DistTh) break;
      if (!(vDistIndex[id].second >> 31)) continue;  // CheckDistEpipolarLine, :928
      const int currentIdx2 = (int)(vDistIndex[id].second & 0x7fffffffu);
      vbMatched2[currentIdx2] = 1;
      match12[idx1] = currentIdx2;
      nmatches++;
      if (check_orientation) {  // :934-944
        float rot = tb->angle1[idx1] - P.angle2[currentIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(idx1);
        else match12[idx1] = -1, nmatches--;  // the reference asserts the range (angles are in [0, 360)); such a match lands in no bin
      }
      break;
    }
  }
  if (check_orientation) {  // :966-984
    int hist[HISTO_LENGTH], ind1, ind2, ind3;
    for (int i = 0; i < HISTO_LENGTH; ++i) hist[i] = (int)rotHist[i].size();
    three_maxima(hist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) {
        match12[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  *n_matches = nmatches;
  return UVO_OK;
}

int uvo_fuse_batch(uvo_matcher* m, int n_targets, const uvo_fuse_target* targets, int nmp, const float* xyz, const float* normal,
                   const float* min_distance_inv, const float* max_distance_inv, const uint8_t* usable, const uint8_t* mp_desc, float th,
                   int32_t* best_idx, int32_t* best_dist) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  if (n_targets < 0 || nmp < 0) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (n_targets == 0 || nmp == 0) return UVO_OK;
  if (!targets || !xyz || !normal || !min_distance_inv || !max_distance_inv || !mp_desc || !best_idx || !best_dist) return matcher_fail(UVO_E_BADARG, "null pointer");
  for (int t = 0; t < n_targets; ++t) {
    const uvo_fuse_target& T = targets[t];
    if (T.n < 0 || T.n > 65535 || T.nlevels < 1 || T.nlevels > 64 || T.max_x <= T.min_x || T.max_y <= T.min_y) return matcher_fail(UVO_E_BADARG, "bad target");
    if (!T.scale_factors || (T.n > 0 && (!T.kp || !T.desc))) return matcher_fail(UVO_E_BADARG, "null pointer in a target");
  }
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // the map points once: xyz | normal | min | max as one float block, usable, descriptors
  std::vector<float> fl((size_t)nmp * 8);
  struct Drain {  // declared after the staging block: runs before it is freed, on every way out
    hipStream_t s;
    bool armed;
    ~Drain() {
      if (armed) (void)hipStreamSynchronize(s);
    }
  } drain{s, true};
  memcpy(&fl[0], xyz, (size_t)nmp * 3 * sizeof(float));
  memcpy(&fl[(size_t)nmp * 3], normal, (size_t)nmp * 3 * sizeof(float));
  memcpy(&fl[(size_t)nmp * 6], min_distance_inv, (size_t)nmp * sizeof(float));
  memcpy(&fl[(size_t)nmp * 7], max_distance_inv, (size_t)nmp * sizeof(float));
  float* d_fl;
  uint8_t *d_usable = nullptr, *d_mpdesc;
  RC(upload(m, B_MISC, fl.data(), fl.size(), &d_fl));
  if (usable) RC(upload(m, B_QPAIR, usable, (size_t)nmp, &d_usable));
  RC(upload(m, B_QDESC, mp_desc, (size_t)nmp * 32, &d_mpdesc));
  // per-point projection outputs (reused by every target) and the results of all targets
  uint8_t* d_work;
  RC(upload<uint8_t>(m, B_CIDX, nullptr, (size_t)nmp * 16, &d_work));
  float* d_u = reinterpret_cast<float*>(d_work);
  float* d_v = d_u + nmp;
  int32_t* d_level = reinterpret_cast<int32_t*>(d_v + nmp);
  uint8_t* d_valid = reinterpret_cast<uint8_t*>(d_level + nmp);
  int32_t* d_best;
  RC(upload<int32_t>(m, B_CAND, nullptr, (size_t)2 * n_targets * nmp, &d_best));
  int max_n = 1, max_lev = 1;
  for (int t = 0; t < n_targets; ++t) max_n = std::max(max_n, targets[t].n), max_lev = std::max(max_lev, targets[t].nlevels);
  // every buffer a target needs is sized for the largest one up front: nothing grows (and synchronises) inside the loop
  uvo_keypoint* d_kp;
  uint8_t* d_desc;
  float* d_sf;
  int32_t* d_cells;
  RC(upload<uvo_keypoint>(m, B_TLEVEL, nullptr, (size_t)max_n, &d_kp));
  RC(upload<uint8_t>(m, B_TDESC, nullptr, (size_t)max_n * 32, &d_desc));
  RC(upload<int32_t>(m, B_START, nullptr, (size_t)64 * 48 + 1 + 2 * (size_t)max_n + 64, &d_cells));
  d_sf = reinterpret_cast<float*>(d_cells + 64 * 48 + 1 + 2 * (size_t)max_n);
  for (int t = 0; t < n_targets; ++t) {
    const uvo_fuse_target& T = targets[t];
    int32_t* bi = d_best + (size_t)t * nmp;
    int32_t* bd = d_best + (size_t)(n_targets + t) * nmp;
    if (T.n == 0) {
      UVO_HIP_CHECK(hipMemsetAsync(bi, 0xff, (size_t)nmp * 4, s));
      UVO_HIP_CHECK(hipMemsetAsync(bd, 0xff, (size_t)nmp * 4, s));
      continue;
    }
    UVO_HIP_CHECK(hipMemcpyAsync(d_kp, T.kp, (size_t)T.n * sizeof(uvo_keypoint), hipMemcpyHostToDevice, s));
    UVO_HIP_CHECK(hipMemcpyAsync(d_desc, T.desc, (size_t)T.n * 32, hipMemcpyHostToDevice, s));
    UVO_HIP_CHECK(hipMemcpyAsync(d_sf, T.scale_factors, (size_t)T.nlevels * sizeof(float), hipMemcpyHostToDevice, s));
    // projection tests of Fuse (:1037-1075) with this target's pose, then the window walk on its grid
    launch_project(s, UVO_PROJECT_FUSE, T.cam, nmp, d_fl, d_fl + (size_t)nmp * 3, d_fl + (size_t)nmp * 6, d_fl + (size_t)nmp * 7, nullptr, d_usable, d_sf,
                   T.nlevels, 0.f, 0.f, d_valid, d_u, d_v, d_level, nullptr);
    launch_fuse_walk(s, d_kp, d_desc, T.n, T.min_x, T.min_y, T.max_x, T.max_y, nmp, d_valid, d_u, d_v, d_level, d_mpdesc, d_sf, th, d_cells,
                     d_cells + 64 * 48 + 1, d_cells + 64 * 48 + 1 + max_n, bi, bd);
  }
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(best_idx, d_best, (size_t)n_targets * nmp * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(best_dist, d_best + (size_t)n_targets * nmp, (size_t)n_targets * nmp * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));  // the only host wait of the batch
  drain.armed = false;
  return UVO_OK;
}

}  // extern "C"
