// Host side of the generic search engine and of the reference-named ORBmatcher entry points built on it
// (include/uvo/uvo.h; kernels in match_engine.hip).  Marshalling only: every decision the reference takes per candidate
// is taken on the device.
#include <algorithm>
#include <cstring>
#include <cmath>
#include <vector>

#include "matcher_priv.hpp"
#include "uvo_math.hpp"

namespace uvo {
void launch_win_count(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y, int nq,
                      const float* d_qx, const float* d_qy, const float* d_qr, const int32_t* d_qmin, const int32_t* d_qmax,
                      const uint8_t* d_qvalid, const uint8_t* d_qdesc, int32_t* d_cell_start, int32_t* d_cell_items, int32_t* d_cell_of_kp,
                      int32_t* d_cand_cnt, int32_t* d_cand_start);
void launch_win_fill(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y, int nq,
                     const float* d_qx, const float* d_qy, const float* d_qr, const int32_t* d_qmin, const int32_t* d_qmax,
                     const uint8_t* d_qvalid, const uint8_t* d_qdesc, const int32_t* d_cell_start, const int32_t* d_cell_items,
                     const int32_t* d_cand_start, uint32_t* d_cand);
void launch_group_dist(hipStream_t s, int nq, int total, const int32_t* d_cand_start, const int32_t* d_cand_idx, const uint8_t* d_qdesc,
                       const uint8_t* d_tdesc, const int32_t* d_tlevel, const float* f12, const float* d_qx, const float* d_qy,
                       const float* d_tx, const float* d_ty, const float* d_sigma2, uint32_t* d_cand);
void launch_match_resolve(hipStream_t s, int nq, int nt, const int32_t* d_cand_start, const uint32_t* d_cand, const uint8_t* d_blocked, int rule,
                          int max_dist, float nn_ratio, int exclusive, int32_t* d_owner, int32_t* d_owner_next, int32_t* d_match,
                          int32_t* d_mdist, int32_t* d_n_matches);
void launch_match_resolve_steal(hipStream_t s, int nq, int nt, const int32_t* d_cand_start, const uint32_t* d_cand, int max_dist, float nn_ratio,
                                int32_t* d_head, int32_t* d_nxt, int32_t* d_tmp, int32_t* d_match, int32_t* d_mdist);
void launch_steal_finalize(hipStream_t s, int nq, const int32_t* d_holder, int32_t* d_match, int32_t* d_mdist, int32_t* d_n_matches);
void launch_rot_filter(hipStream_t s, int nq, const float* d_qangle, const float* d_tangle, int32_t* d_match, int32_t* d_mdist,
                       int32_t* d_n_matches);
void launch_project_sim3(hipStream_t s, const float* r_own, const float* t_own, const float* s_r, const float* t, const uvo_camera_pose& cam, int n,
                         const float* d_xyz, const float* d_min, const float* d_max, const uint8_t* d_usable, const float* d_sf, int nlevels,
                         uint8_t* d_valid, float* d_u, float* d_v, int32_t* d_level);
void launch_project(hipStream_t s, int mode, const uvo_camera_pose& cam, int n, const float* d_xyz, const float* d_normal, const float* d_min,
                    const float* d_max, const float* d_max_raw, const uint8_t* d_usable, const float* d_sf, int nlevels, float log_sf, float cos_limit, uint8_t* d_valid,
                    float* d_u, float* d_v, int32_t* d_level, float* d_cos);
void launch_haloc(hipStream_t s, const float* d_proj, int num_proj, int proj_stride, const uint8_t* d_desc, int n, float* d_hash);
}  // namespace uvo

using namespace uvo;

namespace {

enum Slot {  // uvo_matcher::scratch
  S_KP = 0, S_TDESC, S_BLOCKED, S_QX, S_QY, S_QR, S_QMIN, S_QMAX, S_QVALID, S_QDESC, S_QANGLE, S_TANGLE, S_TLEVEL, S_CELL_START, S_CELL_ITEMS,
  S_CELL_OF, S_CNT, S_START, S_CAND, S_CIDX, S_OWNER, S_OWNER2, S_MATCH, S_MISC
};

// device buffer of at least `bytes` in slot `slot` (contents undefined after growth); nullptr + error code on failure
int ensure(uvo_matcher* m, int slot, size_t bytes, void** out) {
  DevBuf& b = m->scratch[slot];
  if (bytes > b.cap) {
    if (b.p) {
      hipError_t e = hipStreamSynchronize(m->stream);
      if (e != hipSuccess) {
        hip_err_set(e, "hipStreamSynchronize");
        return UVO_E_HIP;
      }
      hipFree(b.p);
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
  if (count && src) {
    hipError_t e = hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, m->stream);
    if (e != hipSuccess) {
      hip_err_set(e, "hipMemcpyAsync");
      return UVO_E_HIP;
    }
  }
  return UVO_OK;
}
template <class T>
int reserve(uvo_matcher* m, int slot, size_t count, T** dev) {
  return upload<T>(m, slot, nullptr, count, dev);
}

#define RC(call)                   \
  do {                             \
    const int _rc = (call);        \
    if (_rc != UVO_OK) return _rc; \
  } while (0)

int check_rule(const uvo_match_rule* r) {
  if (!r) return matcher_fail(UVO_E_BADARG, "null rule");
  if (r->rule < UVO_RULE_BEST_RATIO_SAME_LEVEL || r->rule > UVO_RULE_INIT_STEAL) return matcher_fail(UVO_E_BADARG, "unknown rule");
  if (r->max_dist < 0 || r->max_dist > 256) return matcher_fail(UVO_E_BADARG, "max_dist outside 0..256");
  return UVO_OK;
}

// resolve + optional rotation filter + download; d_cand_start/d_cand hold the packed candidate lists
int finish(uvo_matcher* m, int nq, int nt, const int32_t* d_start, const uint32_t* d_cand, const uint8_t* d_blocked, const uvo_match_rule* rule,
           const float* d_qangle, const float* d_tangle, int32_t* match, int32_t* dist, int* n_matches) {
  hipStream_t s = m->stream;
  int32_t *d_owner, *d_owner2, *d_match;
  RC(reserve(m, S_OWNER, (size_t)nt, &d_owner));
  RC(reserve(m, S_OWNER2, (size_t)nt, &d_owner2));
  RC(reserve(m, S_MATCH, (size_t)2 * nq + 1, &d_match));
  int32_t* d_mdist = d_match + nq;
  int32_t* d_nm = d_match + 2 * nq;
  if (rule->rule == UVO_RULE_INIT_STEAL) {
    // accepts (displaced ones included) -> rotation filter over all of them -> only the queries still holding their target survive
    int32_t* d_tmp;
    RC(reserve(m, S_OWNER2, (size_t)3 * nq + 1, &d_tmp));  // accept list links [nq] + the sweep's scratch [2 nq]
    launch_match_resolve_steal(s, nq, nt, d_start, d_cand, rule->max_dist, rule->nn_ratio, d_owner, d_tmp, d_tmp + nq, d_match, d_mdist);
    if (rule->check_orientation) launch_rot_filter(s, nq, d_qangle, d_tangle, d_match, d_mdist, d_nm);
    launch_steal_finalize(s, nq, d_owner, d_match, d_mdist, d_nm);
  } else {
    launch_match_resolve(s, nq, nt, d_start, d_cand, d_blocked, rule->rule, rule->max_dist, rule->nn_ratio, rule->exclusive ? 1 : 0, d_owner, d_owner2,
                         d_match, d_mdist, d_nm);
    if (rule->check_orientation) launch_rot_filter(s, nq, d_qangle, d_tangle, d_match, d_mdist, d_nm);
  }
  UVO_HIP_CHECK(hipGetLastError());
  int32_t nm = 0;
  UVO_HIP_CHECK(hipMemcpyAsync(match, d_match, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  if (dist) UVO_HIP_CHECK(hipMemcpyAsync(dist, d_mdist, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(&nm, d_nm, 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  if (n_matches) *n_matches = nm;
  return UVO_OK;
}

// merge walk over two feature vectors (std::map iteration with lower_bound jumps == sorted merge, :178-249):
// calls f(group of side 1, group of side 2) for every shared node, ascending node id
template <class F>
void for_shared_nodes(const uvo_feature_vector* a, const uvo_feature_vector* b, F f) {
  int i = 0, j = 0;
  while (i < a->n_nodes && j < b->n_nodes) {
    if (a->node[i] == b->node[j]) {
      f(i, j);
      ++i, ++j;
    } else if (a->node[i] < b->node[j]) {
      ++i;
    } else {
      ++j;
    }
  }
}

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

}  // namespace

extern "C" {

int uvo_match_windows(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, const uint8_t* blocked, int min_x, int min_y,
                      int max_x, int max_y, int nq, const float* qx, const float* qy, const float* qr, const int32_t* qmin_level,
                      const int32_t* qmax_level, const uint8_t* qvalid, const uint8_t* qdesc, const float* qangle, const uvo_match_rule* rule,
                      int32_t* match, int32_t* dist, int* n_matches) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  if (n_matches) *n_matches = 0;
  RC(check_rule(rule));
  if (rule->rule == UVO_RULE_TRIANGULATION) return matcher_fail(UVO_E_BADARG, "the triangulation rule needs caller-given candidates");
  if (n < 0 || nq < 0 || n > 65535 || max_x <= min_x || max_y <= min_y) return matcher_fail(UVO_E_BADARG, "bad sizes (at most 65535 keypoints)");
  if (nq == 0) return UVO_OK;
  if (!match || !qx || !qy || !qr || !qmin_level || !qmax_level || !qvalid || !qdesc) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (rule->check_orientation && !qangle) return matcher_fail(UVO_E_BADARG, "check_orientation needs query angles");
  if (n == 0) {
    for (int i = 0; i < nq; ++i) match[i] = -1;
    if (dist)
      for (int i = 0; i < nq; ++i) dist[i] = -1;
    return UVO_OK;
  }
  if (!kp || !desc) return matcher_fail(UVO_E_BADARG, "null pointer");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  uvo_keypoint* d_kp;
  uint8_t *d_desc, *d_blocked = nullptr, *d_qvalid, *d_qdesc;
  float *d_qx, *d_qy, *d_qr, *d_qangle = nullptr, *d_tangle = nullptr;
  int32_t *d_qmin, *d_qmax, *d_cell_start, *d_cell_items, *d_cell_of, *d_cnt, *d_start;
  RC(upload(m, S_KP, kp, (size_t)n, &d_kp));
  RC(upload(m, S_TDESC, desc, (size_t)n * 32, &d_desc));
  if (blocked) RC(upload(m, S_BLOCKED, blocked, (size_t)n, &d_blocked));
  RC(upload(m, S_QX, qx, (size_t)nq, &d_qx));
  RC(upload(m, S_QY, qy, (size_t)nq, &d_qy));
  RC(upload(m, S_QR, qr, (size_t)nq, &d_qr));
  RC(upload(m, S_QMIN, qmin_level, (size_t)nq, &d_qmin));
  RC(upload(m, S_QMAX, qmax_level, (size_t)nq, &d_qmax));
  RC(upload(m, S_QVALID, qvalid, (size_t)nq, &d_qvalid));
  RC(upload(m, S_QDESC, qdesc, (size_t)nq * 32, &d_qdesc));
  std::vector<float> tangle;
  if (rule->check_orientation) {
    tangle.resize(n);
    for (int k = 0; k < n; ++k) tangle[k] = kp[k].angle;
    RC(upload(m, S_QANGLE, qangle, (size_t)nq, &d_qangle));
    RC(upload(m, S_TANGLE, tangle.data(), (size_t)n, &d_tangle));
  }
  RC(reserve(m, S_CELL_START, (size_t)64 * 48 + 1, &d_cell_start));
  RC(reserve(m, S_CELL_ITEMS, (size_t)n, &d_cell_items));
  RC(reserve(m, S_CELL_OF, (size_t)n, &d_cell_of));
  RC(reserve(m, S_CNT, (size_t)nq + 1, &d_cnt));
  RC(reserve(m, S_START, (size_t)nq + 1, &d_start));
  launch_win_count(s, d_kp, d_desc, n, min_x, min_y, max_x, max_y, nq, d_qx, d_qy, d_qr, d_qmin, d_qmax, d_qvalid, d_qdesc, d_cell_start, d_cell_items,
                   d_cell_of, d_cnt, d_start);
  UVO_HIP_CHECK(hipGetLastError());
  int32_t total = 0;
  UVO_HIP_CHECK(hipMemcpyAsync(&total, d_start + nq, 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));  // also keeps `tangle` alive until its upload has completed
  uint32_t* d_cand;
  RC(reserve(m, S_CAND, (size_t)total, &d_cand));
  launch_win_fill(s, d_kp, d_desc, n, min_x, min_y, max_x, max_y, nq, d_qx, d_qy, d_qr, d_qmin, d_qmax, d_qvalid, d_qdesc, d_cell_start, d_cell_items,
                  d_start, d_cand);
  return finish(m, nq, n, d_start, d_cand, d_blocked, rule, d_qangle, d_tangle, match, dist, n_matches);
}

int uvo_match_groups(uvo_matcher* m, int nq, const uint8_t* qdesc, const float* qangle, int nt, const uint8_t* tdesc, const float* tangle,
                     const int32_t* tlevel, const uint8_t* tblocked, const int32_t* cand_start, const int32_t* cand_idx,
                     const uvo_epipolar* epi, const uvo_match_rule* rule, int32_t* match, int32_t* dist, int* n_matches) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  if (n_matches) *n_matches = 0;
  RC(check_rule(rule));
  if (nq < 0 || nt < 0 || nt > 65535) return matcher_fail(UVO_E_BADARG, "bad sizes (at most 65535 targets)");
  if (nq == 0) return UVO_OK;
  if (!match || !cand_start || !qdesc) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (cand_start[0] != 0) return matcher_fail(UVO_E_BADARG, "cand_start[0] must be 0");
  for (int i = 0; i < nq; ++i)
    if (cand_start[i + 1] < cand_start[i]) return matcher_fail(UVO_E_BADARG, "cand_start must be non-decreasing");
  const int total = cand_start[nq];
  if (total > 0 && (!cand_idx || !tdesc)) return matcher_fail(UVO_E_BADARG, "null pointer");
  for (int e = 0; e < total; ++e)
    if (cand_idx[e] < 0 || cand_idx[e] >= nt) return matcher_fail(UVO_E_BADARG, "candidate index outside the target range");
  if (rule->check_orientation && (!qangle || !tangle)) return matcher_fail(UVO_E_BADARG, "check_orientation needs query and target angles");
  if (rule->rule == UVO_RULE_BEST_RATIO_SAME_LEVEL && !tlevel) return matcher_fail(UVO_E_BADARG, "this rule needs target levels");
  if (rule->rule == UVO_RULE_TRIANGULATION && epi) {
    if (!epi->q_x || !epi->q_y || !epi->t_x || !epi->t_y || !epi->sigma2 || !tlevel || epi->nlevels < 1)
      return matcher_fail(UVO_E_BADARG, "incomplete epipolar description");
    for (int t = 0; t < nt; ++t)
      if (tlevel[t] < 0 || tlevel[t] >= epi->nlevels) return matcher_fail(UVO_E_BADARG, "target level outside the sigma table");
  }
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  uint8_t *d_qdesc, *d_tdesc, *d_blocked = nullptr;
  float *d_qangle = nullptr, *d_tangle = nullptr, *d_qx = nullptr, *d_qy = nullptr, *d_tx = nullptr, *d_ty = nullptr, *d_sigma = nullptr;
  int32_t *d_tlevel = nullptr, *d_start, *d_cidx;
  uint32_t* d_cand;
  RC(upload(m, S_QDESC, qdesc, (size_t)nq * 32, &d_qdesc));
  RC(upload(m, S_TDESC, tdesc, (size_t)nt * 32, &d_tdesc));
  if (tblocked) RC(upload(m, S_BLOCKED, tblocked, (size_t)nt, &d_blocked));
  if (tlevel) RC(upload(m, S_TLEVEL, tlevel, (size_t)nt, &d_tlevel));
  if (rule->check_orientation) {
    RC(upload(m, S_QANGLE, qangle, (size_t)nq, &d_qangle));
    RC(upload(m, S_TANGLE, tangle, (size_t)nt, &d_tangle));
  }
  RC(upload(m, S_START, cand_start, (size_t)nq + 1, &d_start));
  RC(upload(m, S_CIDX, cand_idx, (size_t)total, &d_cidx));
  RC(reserve(m, S_CAND, (size_t)total, &d_cand));
  const bool use_epi = rule->rule == UVO_RULE_TRIANGULATION && epi;
  if (use_epi) {
    RC(upload(m, S_QX, epi->q_x, (size_t)nq, &d_qx));
    RC(upload(m, S_QY, epi->q_y, (size_t)nq, &d_qy));
    RC(upload(m, S_QR, epi->t_x, (size_t)nt, &d_tx));
    RC(upload(m, S_MISC, epi->t_y, (size_t)nt, &d_ty));
    RC(upload(m, S_QMIN, epi->sigma2, (size_t)epi->nlevels, &d_sigma));
  }
  launch_group_dist(s, nq, total, d_start, d_cidx, d_qdesc, d_tdesc, d_tlevel, use_epi ? epi->f12 : nullptr, d_qx, d_qy, d_tx, d_ty, d_sigma, d_cand);
  return finish(m, nq, nt, d_start, d_cand, d_blocked, rule, d_qangle, d_tangle, match, dist, n_matches);
}

int uvo_search_by_projection_kf(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y,
                                int32_t* assigned, int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid,
                                const uint8_t* mp_desc, const float* kf_angle, const float* scale_factors, int nlevels, float th, int orb_dist,
                                int check_orientation, int* n_matches) {
  if (!m || !n_matches) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_matches = 0;
  if (n < 0 || nmp < 0 || nlevels < 1) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (n == 0 || nmp == 0) return UVO_OK;
  if (!assigned || !u || !v || !level || !valid || !mp_desc || !scale_factors) return matcher_fail(UVO_E_BADARG, "null pointer");
  std::vector<float> r(nmp, 0.f);
  std::vector<int32_t> lo(nmp, 0), hi(nmp, 0), match(nmp, -1);
  std::vector<uint8_t> blocked(n);
  for (int i = 0; i < nmp; ++i) {
    if (!valid[i]) continue;
    if (level[i] < 0 || level[i] >= nlevels) return matcher_fail(UVO_E_BADARG, "map point level outside 0..nlevels-1");
    r[i] = th * scale_factors[level[i]];                // :1672
    lo[i] = level[i] - 1, hi[i] = level[i] + 1;         // :1674
  }
  for (int k = 0; k < n; ++k) blocked[k] = assigned[k] >= 0;  // :1690
  uvo_match_rule rule{UVO_RULE_BEST_ONLY, orb_dist > 256 ? 256 : orb_dist, 0.f, 1, check_orientation};
  int rc = uvo_match_windows(m, kp, n, desc, blocked.data(), min_x, min_y, max_x, max_y, nmp, u, v, r.data(), lo.data(), hi.data(), valid, mp_desc,
                             kf_angle, &rule, match.data(), nullptr, n_matches);
  if (rc) return rc;
  for (int i = 0; i < nmp; ++i)
    if (match[i] >= 0) assigned[match[i]] = i;
  return UVO_OK;
}

int uvo_fuse(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y, int nmp,
             const float* u, const float* v, const int32_t* level, const uint8_t* valid, const uint8_t* mp_desc, const float* scale_factors,
             int nlevels, float th, int32_t* best_idx, int32_t* best_dist) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  if (n < 0 || nmp < 0 || nlevels < 1) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (nmp == 0) return UVO_OK;
  if (!best_idx || !best_dist || !u || !v || !level || !valid || !mp_desc || !scale_factors) return matcher_fail(UVO_E_BADARG, "null pointer");
  std::vector<float> r(nmp, 0.f);
  std::vector<int32_t> lo(nmp, 0), hi(nmp, 0);
  for (int i = 0; i < nmp; ++i) {
    if (!valid[i]) continue;
    if (level[i] < 0 || level[i] >= nlevels) return matcher_fail(UVO_E_BADARG, "map point level outside 0..nlevels-1");
    r[i] = th * scale_factors[level[i]];         // :1077
    lo[i] = level[i] - 1, hi[i] = level[i];      // :1094
  }
  uvo_match_rule rule{UVO_RULE_BEST_ONLY, 50 /* TH_LOW :41 */, 0.f, 0, 0};
  int nm = 0;
  return uvo_match_windows(m, kp, n, desc, nullptr, min_x, min_y, max_x, max_y, nmp, u, v, r.data(), lo.data(), hi.data(), valid, mp_desc, nullptr,
                           &rule, best_idx, best_dist, &nm);
}

int uvo_search_by_bow(uvo_matcher* m, int kf_kf, const uvo_feature_vector* fv1, int n1, const uint8_t* desc1, const float* angle1,
                      const uint8_t* usable1, const uvo_feature_vector* fv2, int n2, const uint8_t* desc2, const float* angle2,
                      const uint8_t* usable2, float nnratio, int check_orientation, int32_t* match12, int* n_matches) {
  if (!m || !n_matches) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_matches = 0;
  if (n1 < 0 || n2 < 0) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (n1 == 0) return UVO_OK;
  if (!match12 || !usable1) return matcher_fail(UVO_E_BADARG, "null pointer");
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  RC(check_fv(fv1, n1));
  RC(check_fv(fv2, n2));
  if (n2 == 0) return UVO_OK;
  // queries in the reference's visiting order: shared nodes ascending, features of side 1 in node order (:181-186 / :744-748)
  std::vector<int32_t> q_of, start(1, 0), cidx;
  for_shared_nodes(fv1, fv2, [&](int a, int b) {
    for (int e = fv1->start[a]; e < fv1->start[a + 1]; ++e) {
      const int idx1 = fv1->feat[e];
      if (!usable1[idx1]) continue;
      q_of.push_back(idx1);
      cidx.insert(cidx.end(), fv2->feat + fv2->start[b], fv2->feat + fv2->start[b + 1]);
      start.push_back((int32_t)cidx.size());
    }
  });
  const int nq = (int)q_of.size();
  if (nq == 0) return UVO_OK;
  std::vector<uint8_t> qdesc((size_t)nq * 32), blocked;
  std::vector<float> qangle(nq, 0.f);
  for (int i = 0; i < nq; ++i) {
    memcpy(&qdesc[(size_t)i * 32], desc1 + (size_t)q_of[i] * 32, 32);
    if (angle1) qangle[i] = angle1[q_of[i]];
  }
  if (usable2) {
    blocked.resize(n2);
    for (int k = 0; k < n2; ++k) blocked[k] = !usable2[k];
  }
  uvo_match_rule rule{kf_kf ? UVO_RULE_BEST_RATIO_LT : UVO_RULE_BEST_RATIO_LE, 50 /* TH_LOW */, nnratio, 1, check_orientation};
  std::vector<int32_t> match(nq, -1);
  int rc = uvo_match_groups(m, nq, qdesc.data(), angle1 ? qangle.data() : nullptr, n2, desc2, angle2, nullptr, usable2 ? blocked.data() : nullptr,
                            start.data(), cidx.data(), nullptr, &rule, match.data(), nullptr, n_matches);
  if (rc) return rc;
  for (int i = 0; i < nq; ++i) match12[q_of[i]] = match[i];
  return UVO_OK;
}

int uvo_search_for_triangulation(uvo_matcher* m, const uvo_feature_vector* fv1, const uvo_keypoint* kp1, int n1, const uint8_t* desc1,
                                 const uint8_t* has_mp1, const uvo_feature_vector* fv2, const uvo_keypoint* kp2, int n2, const uint8_t* desc2,
                                 const uint8_t* has_mp2, const float* f12, const float* sigma2, int nlevels, int check_orientation,
                                 int32_t* match12, int* n_matches) {
  if (!m || !n_matches) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_matches = 0;
  if (n1 < 0 || n2 < 0 || nlevels < 1) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (n1 == 0) return UVO_OK;
  if (!match12 || !kp1 || !desc1 || !has_mp1 || !f12 || !sigma2) return matcher_fail(UVO_E_BADARG, "null pointer");
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  RC(check_fv(fv1, n1));
  RC(check_fv(fv2, n2));
  if (n2 == 0) return UVO_OK;
  if (!kp2 || !desc2 || !has_mp2) return matcher_fail(UVO_E_BADARG, "null pointer");
  std::vector<int32_t> q_of, start(1, 0), cidx;
  for_shared_nodes(fv1, fv2, [&](int a, int b) {
    for (int e = fv1->start[a]; e < fv1->start[a + 1]; ++e) {
      const int idx1 = fv1->feat[e];
      if (has_mp1[idx1]) continue;  // :885-889
      q_of.push_back(idx1);
      cidx.insert(cidx.end(), fv2->feat + fv2->start[b], fv2->feat + fv2->start[b + 1]);
      start.push_back((int32_t)cidx.size());
    }
  });
  const int nq = (int)q_of.size();
  if (nq == 0) return UVO_OK;
  std::vector<uint8_t> qdesc((size_t)nq * 32);
  std::vector<float> qangle(nq), qx(nq), qy(nq), tx(n2), ty(n2), tangle(n2);
  std::vector<int32_t> tlevel(n2);
  for (int i = 0; i < nq; ++i) {
    memcpy(&qdesc[(size_t)i * 32], desc1 + (size_t)q_of[i] * 32, 32);
    qangle[i] = kp1[q_of[i]].angle, qx[i] = kp1[q_of[i]].x, qy[i] = kp1[q_of[i]].y;
  }
  for (int k = 0; k < n2; ++k) tx[k] = kp2[k].x, ty[k] = kp2[k].y, tangle[k] = kp2[k].angle, tlevel[k] = kp2[k].octave;
  uvo_epipolar epi;
  memcpy(epi.f12, f12, sizeof(epi.f12));
  epi.q_x = qx.data(), epi.q_y = qy.data(), epi.t_x = tx.data(), epi.t_y = ty.data(), epi.sigma2 = sigma2, epi.nlevels = nlevels;
  uvo_match_rule rule{UVO_RULE_TRIANGULATION, 50 /* TH_LOW */, 0.f, 1, check_orientation};
  std::vector<int32_t> match(nq, -1);
  int rc = uvo_match_groups(m, nq, qdesc.data(), qangle.data(), n2, desc2, tangle.data(), tlevel.data(), has_mp2, start.data(), cidx.data(), &epi,
                            &rule, match.data(), nullptr, n_matches);
  if (rc) return rc;
  for (int i = 0; i < nq; ++i) match12[q_of[i]] = match[i];
  return UVO_OK;
}

int uvo_project_points(uvo_matcher* m, int mode, const uvo_camera_pose* cam, int npts, const float* xyz, const float* normal,
                       const float* min_distance_inv, const float* max_distance_inv, const float* max_distance, const uint8_t* usable,
                       const float* scale_factors, int nlevels, float scale_factor, float viewing_cos_limit, uint8_t* valid, float* u, float* v,
                       int32_t* level, float* view_cos) {
  if (!m || !cam) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (mode < UVO_PROJECT_FRUSTUM || mode > UVO_PROJECT_PIXEL) return matcher_fail(UVO_E_BADARG, "unknown projection mode");
  const bool pixel_only = mode == UVO_PROJECT_PIXEL || mode == UVO_PROJECT_PIXEL_BOUNDED;  // u, v and at most the image-bounds test
  if (npts < 0 || nlevels < 1 || nlevels > 64) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (npts == 0) return UVO_OK;
  if (!xyz || !scale_factors || !valid || !u || !v || !level) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (!pixel_only && !min_distance_inv) return matcher_fail(UVO_E_BADARG, "this mode needs the minimum invariance distance");
  if (!pixel_only && mode != UVO_PROJECT_KF_RELOC && !max_distance_inv) return matcher_fail(UVO_E_BADARG, "this mode needs the maximum invariance distance");
  if (mode == UVO_PROJECT_FRUSTUM && !max_distance) return matcher_fail(UVO_E_BADARG, "PredictScale needs the raw mfMaxDistance");
  if (!pixel_only && mode != UVO_PROJECT_KF_RELOC && !normal) return matcher_fail(UVO_E_BADARG, "this mode needs the point normals");
  if (mode == UVO_PROJECT_FRUSTUM && !(scale_factor > 1.0f)) return matcher_fail(UVO_E_BADARG, "scale_factor must be > 1");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  float *d_xyz, *d_normal = nullptr, *d_min = nullptr, *d_max = nullptr, *d_max_raw = nullptr, *d_sf, *d_u;
  uint8_t *d_usable = nullptr, *d_valid;
  RC(upload(m, S_QX, xyz, (size_t)npts * 3, &d_xyz));
  if (normal) RC(upload(m, S_QY, normal, (size_t)npts * 3, &d_normal));
  if (min_distance_inv) RC(upload(m, S_QR, min_distance_inv, (size_t)npts, &d_min));
  if (max_distance_inv) RC(upload(m, S_QANGLE, max_distance_inv, (size_t)npts, &d_max));
  if (max_distance) RC(upload(m, S_MISC, max_distance, (size_t)npts, &d_max_raw));
  if (usable) RC(upload(m, S_QVALID, usable, (size_t)npts, &d_usable));
  RC(upload(m, S_TANGLE, scale_factors, (size_t)nlevels, &d_sf));
  RC(reserve(m, S_MATCH, (size_t)npts * 4, &d_u));  // u, v, level, view_cos
  RC(reserve(m, S_BLOCKED, (size_t)npts, &d_valid));
  float* d_v = d_u + npts;
  int32_t* d_level = reinterpret_cast<int32_t*>(d_u + 2 * (size_t)npts);
  float* d_cos = d_u + 3 * (size_t)npts;
  const float log_sf = uvo_logf(scale_factor);  // mfLogScaleFactor = log(mfScaleFactor), src/FrameKTL.cc:97 (as intended)
  launch_project(s, mode, *cam, npts, d_xyz, d_normal, d_min, d_max, d_max_raw, d_usable, d_sf, nlevels, log_sf, viewing_cos_limit, d_valid, d_u, d_v, d_level,
                 d_cos);
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(valid, d_valid, (size_t)npts, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(u, d_u, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(v, d_v, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(level, d_level, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  if (view_cos) UVO_HIP_CHECK(hipMemcpyAsync(view_cos, d_cos, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  return UVO_OK;
}

// src/ORBmatcher.cc:299-303 (= :1145-1149).  cv::Mat::dot accumulates the three products in double; `sRcw / scw` is
// convertTo(alpha = 1./scw), whose 32F kernel multiplies by (float)alpha; `-Rcw.t() * tcw` is gemm(GEMM_1_T, alpha = -1): general
// path, double accumulation.
int uvo_sim3_decompose(const float* scw_mat, int row_stride, uvo_camera_pose* cam) {
  if (!scw_mat || !cam || row_stride < 4) return matcher_fail(UVO_E_BADARG, "null pointer / row_stride < 4");
  double dot = 0.0;
  for (int k = 0; k < 3; ++k) dot += (double)scw_mat[k] * (double)scw_mat[k];
  const float scw = (float)std::sqrt(dot);
  if (!(scw > 0.f)) return matcher_fail(UVO_E_BADARG, "Scw has a zero first row");
  const float inv = (float)(1.0 / (double)scw);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) cam->rcw[3 * i + j] = scw_mat[i * row_stride + j] * inv;
    cam->tcw[i] = scw_mat[i * row_stride + 3] * inv;
  }
  for (int c = 0; c < 3; ++c) {
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) acc += (double)cam->rcw[3 * k + c] * (double)cam->tcw[k];
    cam->ow[c] = (float)(acc * -1.0);
  }
  return UVO_OK;
}

// src/ORBmatcher.cc:1284-1287: `s12*R12` and `(1.0/s12)*R12.t()` are convertTo with (float)alpha; `-sR21*t12` is the small-matrix
// gemm: fp32 row sum, times alpha = -1 in double
int uvo_sim3_relative(float s12, const float* r12, const float* t12, float* s_r12, float* s_r21, float* t21) {
  if (!r12 || !t12 || !s_r12 || !s_r21 || !t21) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (!(s12 > 0.f)) return matcher_fail(UVO_E_BADARG, "s12 must be positive");
  for (int i = 0; i < 9; ++i) s_r12[i] = r12[i] * s12;
  const float inv = (float)(1.0 / (double)s12);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) s_r21[3 * i + j] = r12[3 * j + i] * inv;
  for (int i = 0; i < 3; ++i) {
    const float t0 = s_r21[3 * i] * t12[0] + s_r21[3 * i + 1] * t12[1] + s_r21[3 * i + 2] * t12[2];
    t21[i] = (float)((double)t0 * -1.0);
  }
  return UVO_OK;
}

int uvo_project_sim3(uvo_matcher* m, const float* r_own, const float* t_own, const float* s_r, const float* t, const uvo_camera_pose* cam_other,
                     int npts, const float* xyz, const float* min_distance_inv, const float* max_distance_inv, const uint8_t* usable,
                     const float* scale_factors, int nlevels, uint8_t* valid, float* u, float* v, int32_t* level) {
  if (!m || !r_own || !t_own || !s_r || !t || !cam_other) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (npts < 0 || nlevels < 1 || nlevels > 64) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (npts == 0) return UVO_OK;
  if (!xyz || !min_distance_inv || !max_distance_inv || !scale_factors || !valid || !u || !v || !level)
    return matcher_fail(UVO_E_BADARG, "null pointer");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  float *d_xyz, *d_min, *d_max, *d_sf, *d_u;
  uint8_t *d_usable = nullptr, *d_valid;
  RC(upload(m, S_QX, xyz, (size_t)npts * 3, &d_xyz));
  RC(upload(m, S_QR, min_distance_inv, (size_t)npts, &d_min));
  RC(upload(m, S_QANGLE, max_distance_inv, (size_t)npts, &d_max));
  if (usable) RC(upload(m, S_QVALID, usable, (size_t)npts, &d_usable));
  RC(upload(m, S_TANGLE, scale_factors, (size_t)nlevels, &d_sf));
  RC(reserve(m, S_MATCH, (size_t)npts * 3, &d_u));  // u, v, level
  RC(reserve(m, S_BLOCKED, (size_t)npts, &d_valid));
  float* d_v = d_u + npts;
  int32_t* d_level = reinterpret_cast<int32_t*>(d_u + 2 * (size_t)npts);
  launch_project_sim3(s, r_own, t_own, s_r, t, *cam_other, npts, d_xyz, d_min, d_max, d_usable, d_sf, nlevels, d_valid, d_u, d_v, d_level);
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(valid, d_valid, (size_t)npts, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(u, d_u, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(v, d_v, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(level, d_level, (size_t)npts * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  return UVO_OK;
}

int uvo_search_by_projection_sim3(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y,
                                  int32_t* matched, int nmp, const float* u, const float* v, const int32_t* level, const uint8_t* valid,
                                  const uint8_t* mp_desc, const float* scale_factors, int nlevels, int th, int* n_matches) {
  if (!m || !n_matches) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_matches = 0;
  if (n < 0 || nmp < 0 || nlevels < 1) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (n == 0 || nmp == 0) return UVO_OK;
  if (!matched || !u || !v || !level || !valid || !mp_desc || !scale_factors) return matcher_fail(UVO_E_BADARG, "null pointer");
  std::vector<float> r(nmp, 0.f);
  std::vector<int32_t> lo(nmp, 0), hi(nmp, 0), match(nmp, -1);
  std::vector<uint8_t> blocked(n);
  for (int i = 0; i < nmp; ++i) {
    if (!valid[i]) continue;
    if (level[i] < 0 || level[i] >= nlevels) return matcher_fail(UVO_E_BADARG, "map point level outside 0..nlevels-1");
    r[i] = th * scale_factors[level[i]];      // :357 (int th promoted to float)
    lo[i] = level[i] - 1, hi[i] = level[i];   // :377
  }
  for (int k = 0; k < n; ++k) blocked[k] = matched[k] >= 0;  // :372
  uvo_match_rule rule{UVO_RULE_BEST_ONLY, 50 /* TH_LOW :41 */, 0.f, 1, 0};
  int rc = uvo_match_windows(m, kp, n, desc, blocked.data(), min_x, min_y, max_x, max_y, nmp, u, v, r.data(), lo.data(), hi.data(), valid, mp_desc,
                             nullptr, &rule, match.data(), nullptr, n_matches);
  if (rc) return rc;
  for (int i = 0; i < nmp; ++i)
    if (match[i] >= 0) matched[match[i]] = i;
  return UVO_OK;
}

int uvo_search_by_sim3(uvo_matcher* m, const uvo_keypoint* kp1, int n1, const uint8_t* desc1, const int32_t* bounds1, const uvo_keypoint* kp2,
                       int n2, const uint8_t* desc2, const int32_t* bounds2, const float* u12, const float* v12, const int32_t* level12,
                       const uint8_t* valid12, const uint8_t* mp_desc1, const float* u21, const float* v21, const int32_t* level21,
                       const uint8_t* valid21, const uint8_t* mp_desc2, const float* scale_factors1, int nlevels1, const float* scale_factors2,
                       int nlevels2, float th, int32_t* match12, int* n_found) {
  if (!m || !n_found) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_found = 0;
  if (n1 < 0 || n2 < 0 || nlevels1 < 1 || nlevels2 < 1) return matcher_fail(UVO_E_BADARG, "bad sizes");
  if (n1 == 0) return UVO_OK;
  if (!match12) return matcher_fail(UVO_E_BADARG, "null pointer");
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  if (n2 == 0) return UVO_OK;
  if (!bounds1 || !bounds2 || !u12 || !v12 || !level12 || !valid12 || !mp_desc1 || !u21 || !v21 || !level21 || !valid21 || !mp_desc2 ||
      !scale_factors1 || !scale_factors2)
    return matcher_fail(UVO_E_BADARG, "null pointer");
  // one direction: queries = the projected map points, targets = the other key frame's key points (:1361-1394 / :1443-1476)
  auto direction = [&](const uvo_keypoint* kp, int n, const uint8_t* desc, const int32_t* b, int nq, const float* u, const float* v,
                       const int32_t* level, const uint8_t* valid, const uint8_t* qdesc, const float* sf, int nl, std::vector<int32_t>& out) -> int {
    std::vector<float> r(nq, 0.f);
    std::vector<int32_t> lo(nq, 0), hi(nq, 0);
    for (int i = 0; i < nq; ++i) {
      if (!valid[i]) continue;
      if (level[i] < 0 || level[i] >= nl) return matcher_fail(UVO_E_BADARG, "map point level outside 0..nlevels-1");
      r[i] = th * sf[level[i]];
      lo[i] = level[i] - 1, hi[i] = level[i];
    }
    out.assign(nq, -1);
    uvo_match_rule rule{UVO_RULE_BEST_ONLY, 100 /* TH_HIGH :40 */, 0.f, 0, 0};
    int nm = 0;
    return uvo_match_windows(m, kp, n, desc, nullptr, b[0], b[1], b[2], b[3], nq, u, v, r.data(), lo.data(), hi.data(), valid, qdesc, nullptr, &rule,
                             out.data(), nullptr, &nm);
  };
  std::vector<int32_t> vnMatch1, vnMatch2;
  RC(direction(kp2, n2, desc2, bounds2, n1, u12, v12, level12, valid12, mp_desc1, scale_factors2, nlevels2, vnMatch1));
  RC(direction(kp1, n1, desc1, bounds1, n2, u21, v21, level21, valid21, mp_desc2, scale_factors1, nlevels1, vnMatch2));
  int found = 0;
  for (int i1 = 0; i1 < n1; ++i1) {  // :1479-1494
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0 && vnMatch2[idx2] == i1) match12[i1] = idx2, ++found;
  }
  *n_found = found;
  return UVO_OK;
}

int uvo_haloc_hash(uvo_matcher* m, const float* proj, int num_proj, int proj_stride, const uint8_t* desc, int n, float* hash) {
  if (!m || !hash) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (num_proj < 1 || n < 0 || proj_stride < n) return matcher_fail(UVO_E_BADARG, "bad sizes (proj_stride must cover the descriptor rows)");
  if (n == 0) {  // :66 the zero-initialised histogram
    for (int k = 0; k < num_proj * 32; ++k) hash[k] = 0.0f;
    return UVO_OK;
  }
  if (!proj || !desc) return matcher_fail(UVO_E_BADARG, "null pointer");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  float *d_proj, *d_hash;
  uint8_t* d_desc;
  RC(upload(m, S_QX, proj, (size_t)num_proj * proj_stride, &d_proj));
  RC(upload(m, S_QDESC, desc, (size_t)n * 32, &d_desc));
  RC(reserve(m, S_QY, (size_t)num_proj * 32, &d_hash));
  launch_haloc(s, d_proj, num_proj, proj_stride, d_desc, n, d_hash);
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(hash, d_hash, (size_t)num_proj * 32 * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  return UVO_OK;
}

}  // extern "C"
