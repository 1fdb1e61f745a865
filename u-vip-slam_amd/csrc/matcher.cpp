// Host side of the matcher entry points of include/uvo/uvo.h (replacing the arithmetic and search cores of
// USLAM::ORBmatcher, src/ORBmatcher.cc, and Utils::ratioMatching, include/utils.h:81-111).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "matcher_priv.hpp"
#include "uvo_math.hpp"

namespace uvo {
void launch_sbp(hipStream_t s, const uvo_keypoint* d_kp, int n, const uint8_t* d_desc, int min_x, int min_y, int max_x, int max_y,
                int32_t* d_assigned, int nmp, const float* d_px, const float* d_py, const int32_t* d_level, const float* d_vc,
                const uint8_t* d_inview, const uint8_t* d_mpdesc, const float* d_scale, float th, float nnratio, int32_t* d_cell_start,
                int32_t* d_cell_items, int32_t* d_cell_of_kp, int32_t* d_cand_cnt, int32_t* d_cand_start, uint32_t* d_cand, int32_t* d_owner,
                int32_t* d_owner_next, int32_t* d_choice, int32_t* d_n_matches, int stage, int64_t cand_cap);
void launch_project(hipStream_t s, int mode, const uvo_camera_pose& cam, int n, const float* d_xyz, const float* d_normal, const float* d_min,
                    const float* d_max, const float* d_max_raw, const uint8_t* d_usable, const float* d_sf, int nlevels, float log_sf, float cos_limit,
                    uint8_t* d_valid, float* d_u, float* d_v, int32_t* d_level, float* d_cos);
int matcher_fail(int code, const char* msg);
}  // namespace uvo
extern "C" hipStream_t uvo_extractor_stream_internal(uvo_extractor* h);
extern "C" int uvo_extractor_device_internal(uvo_extractor* h);
extern "C" void uvo_extractor_add_follower_internal(uvo_extractor* h, uvo_matcher* m);
extern "C" void uvo_extractor_drop_follower_internal(uvo_extractor* h, uvo_matcher* m);

using namespace uvo;

namespace uvo {
int matcher_fail(int code, const char* msg) { return fail(code, msg); }
}  // namespace uvo

extern "C" {

int uvo_matcher_create(const uvo_matcher_cfg* cfg, uvo_matcher** out) {
  if (!cfg || !out) return matcher_fail(UVO_E_BADARG, "null pointer");
  *out = nullptr;
  if (cfg->max_query < 1 || cfg->max_train < 1 || cfg->max_batch < 1 || cfg->max_map_points < 0 || cfg->max_query > 65535 ||
      cfg->max_train > 65535)
    return matcher_fail(UVO_E_BADARG, "bad matcher configuration");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return matcher_fail(UVO_E_NODEVICE, "no HIP device available (no CPU fallback exists)");
  if (cfg->device < 0 || cfg->device >= ndev) return matcher_fail(UVO_E_BADARG, "device ordinal out of range");
  uvo_matcher* m = new uvo_matcher();
  m->cfg = *cfg;
  m->device = cfg->device;
  if (hipSetDevice(m->device) != hipSuccess || hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete m;
    return matcher_fail(UVO_E_HIP, "stream creation failed");
  }
  m->stream = m->own_stream;
  const size_t B = cfg->max_batch, Q = cfg->max_query, T = cfg->max_train, MP = std::max(cfg->max_map_points, 1);
  int rc;
#define A(call)                  \
  if ((rc = (call)) != UVO_OK) { \
    uvo_matcher_destroy(m);      \
    return rc;                   \
  }
  A(m_alloc(&m->d_q, Q * 32));
  A(m_alloc(&m->d_t, T * 32));
  A(m_alloc(&m->d_idx0, B * Q));
  A(m_alloc(&m->d_idx1, B * Q));
  A(m_alloc(&m->d_d0, B * Q));
  A(m_alloc(&m->d_d1, B * Q));
  A(m_alloc(&m->d_kp, Q));
  A(m_alloc(&m->d_assigned, Q));
  A(m_alloc(&m->d_cell_start, (size_t)64 * 48 + 1));
  A(m_alloc(&m->d_cell_items, Q));
  A(m_alloc(&m->d_cell_of_kp, Q));
  A(m_alloc(&m->d_owner, Q));
  A(m_alloc(&m->d_owner_next, Q));
  A(m_alloc(&m->d_px, MP));
  A(m_alloc(&m->d_py, MP));
  A(m_alloc(&m->d_vc, MP));
  A(m_alloc(&m->d_level, MP));
  A(m_alloc(&m->d_inview, MP));
  A(m_alloc(&m->d_mpdesc, MP * 32));
  A(m_alloc(&m->d_cand_cnt, MP + 1));
  A(m_alloc(&m->d_cand_start, MP + 1));
  A(m_alloc(&m->d_choice, MP));
  A(m_alloc(&m->d_scale, (size_t)kMaxLevels * 4));
  A(m_alloc(&m->d_nm, (size_t)1));
#undef A
  if (hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess) {
    uvo_matcher_destroy(m);
    return matcher_fail(UVO_E_HIP, "event creation failed");
  }
  *out = m;
  return UVO_OK;
}

void uvo_matcher_destroy(uvo_matcher* m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->attached_to) uvo_extractor_drop_follower_internal(m->attached_to, m);
  if (m->stream) hipStreamSynchronize(m->stream);
  if (m->own_stream && m->own_stream != m->stream) hipStreamSynchronize(m->own_stream);
  void* ptrs[] = {m->d_q,      m->d_t,     m->d_mask,       m->d_idx0,       m->d_idx1,       m->d_d0,    m->d_d1,         m->d_dist,  m->d_kp,
                  m->d_px,     m->d_py,    m->d_vc,         m->d_scale,      m->d_level,      m->d_assigned, m->d_cell_start, m->d_cell_items,
                  m->d_cell_of_kp, m->d_cand_cnt, m->d_cand_start, m->d_owner, m->d_owner_next, m->d_choice, m->d_nm, m->d_inview, m->d_mpdesc,
                  m->d_cand, m->d_md, m->d_moff, m->d_mres};
  for (void* p : ptrs)
    if (p) hipFree(p);
  for (DevBuf& b : m->scratch)
    if (b.p) hipFree(b.p);
  uvo::tri_batch_free(m->tri_batch);
  if (m->d_arena) hipFree(m->d_arena);
  if (m->h_arena) (void)hipHostFree(m->h_arena);
  m->prof.clear();
  if (m->ev) (void)hipEventDestroy(m->ev);
  if (m->own_stream) hipStreamDestroy(m->own_stream);
  delete m;
}

int uvo_matcher_synchronize(uvo_matcher* m) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  UVO_HIP_CHECK(hipStreamSynchronize(m->stream));
  return UVO_OK;
}

int uvo_hamming_knn2(uvo_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* mask, int32_t* idx0, uint16_t* d0,
                     int32_t* idx1, uint16_t* d1) {
  if (!m || !idx0 || !d0 || !idx1 || !d1) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (nq < 0 || nt < 0 || nq > m->cfg.max_query || nt > m->cfg.max_train) return matcher_fail(UVO_E_BADARG, "descriptor count outside handle capacity");
  if (nq == 0) return UVO_OK;  // ratioMatching returns early on empty inputs (include/utils.h:85-86)
  if (!q || (nt > 0 && !t)) return matcher_fail(UVO_E_BADARG, "null descriptor pointer");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
  if (nt > 0) UVO_HIP_CHECK(hipMemcpyAsync(m->d_t, t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
  const uint8_t* dmask = nullptr;
  if (mask && nt > 0) {
    const size_t mb = (size_t)nq * nt;
    if (mb > m->mask_bytes) {
      UVO_HIP_CHECK(hipStreamSynchronize(s));
      if (m->d_mask) hipFree(m->d_mask);
      m->d_mask = nullptr;
      int rc = m_alloc(&m->d_mask, mb);
      if (rc) return rc;
      m->mask_bytes = mb;
    }
    UVO_HIP_CHECK(hipMemcpyAsync(m->d_mask, mask, mb, hipMemcpyHostToDevice, s));
    dmask = m->d_mask;
  }
  launch_knn2(s, 1, nq, m->d_q, nullptr, nq, 0, m->d_t, nullptr, nt, 0, dmask, m->cfg.max_query, m->d_idx0, m->d_d0, m->d_idx1, m->d_d1);
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(idx0, m->d_idx0, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(idx1, m->d_idx1, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(d0, m->d_d0, (size_t)nq * 2, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(d1, m->d_d1, (size_t)nq * 2, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  return UVO_OK;
}

int uvo_hamming_knn2_batch_device(uvo_matcher* m, int pairs, const uint8_t* d_q, const int32_t* d_nq, int q_stride, const uint8_t* d_t,
                                  const int32_t* d_nt, int t_stride, int32_t* d_idx0, uint16_t* d_d0, int32_t* d_idx1, uint16_t* d_d1) {
  if (!m || !d_q || !d_t || !d_nq || !d_nt || !d_idx0 || !d_d0 || !d_idx1 || !d_d1) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (pairs < 1 || pairs > m->cfg.max_batch || q_stride < 1 || t_stride < 1) return matcher_fail(UVO_E_BADARG, "bad batch / stride");
  // the launch covers max_query rows per pair and the outputs are [pairs][max_query]: a wider query slice would lose rows silently
  if (q_stride > m->cfg.max_query) return matcher_fail(UVO_E_BADARG, "q_stride above the handle's max_query");
  if (t_stride > 65535) return matcher_fail(UVO_E_BADARG, "t_stride above 65535 (train indices are packed in 16 bits)");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  {
    Profiler::Scope ps(&m->prof, "k_knn2", m->stream);
    launch_knn2(m->stream, pairs, m->cfg.max_query, d_q, d_nq, 0, q_stride, d_t, d_nt, 0, t_stride, nullptr, m->cfg.max_query, d_idx0, d_d0,
                d_idx1, d_d1);
  }
  UVO_HIP_CHECK(hipGetLastError());
  return UVO_OK;
}

int uvo_hamming_matrix(uvo_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* dist) {
  if (!m || !dist) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (nq < 0 || nt < 0 || nq > m->cfg.max_query || nt > m->cfg.max_train) return matcher_fail(UVO_E_BADARG, "descriptor count outside handle capacity");
  if (nq == 0 || nt == 0) return UVO_OK;
  if (!q || !t) return matcher_fail(UVO_E_BADARG, "null descriptor pointer");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  const size_t need = (size_t)nq * nt;
  if (need > m->dist_elems) {
    UVO_HIP_CHECK(hipStreamSynchronize(s));
    if (m->d_dist) hipFree(m->d_dist);
    m->d_dist = nullptr;
    int rc = m_alloc(&m->d_dist, need);
    if (rc) return rc;
    m->dist_elems = need;
  }
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_t, t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
  launch_matrix(s, m->d_q, nq, m->d_t, nt, m->d_dist);
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(dist, m->d_dist, need * 2, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  return UVO_OK;
}

int uvo_distinctive_descriptors(uvo_matcher* m, const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx,
                                int32_t* best_median) {
  if (!m || !offsets || !best_idx || !best_median) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (npoints < 0) return matcher_fail(UVO_E_BADARG, "negative point count");
  if (npoints == 0) return UVO_OK;
  if (offsets[0] != 0) return matcher_fail(UVO_E_BADARG, "offsets[0] must be 0");
  for (int p = 0; p < npoints; ++p)
    if (offsets[p + 1] < offsets[p] || offsets[p + 1] - offsets[p] > 65535) return matcher_fail(UVO_E_BADARG, "offsets must be non-decreasing, <= 65535 rows per point");
  const size_t rows = (size_t)offsets[npoints];
  if (rows > 0 && !desc) return matcher_fail(UVO_E_BADARG, "null descriptor pointer");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  if (rows > m->md_rows || (size_t)npoints > m->md_points) {
    UVO_HIP_CHECK(hipStreamSynchronize(s));
    for (void* p : {(void*)m->d_md, (void*)m->d_moff, (void*)m->d_mres})
      if (p) hipFree(p);
    m->d_md = nullptr, m->d_moff = nullptr, m->d_mres = nullptr;
    const size_t r2 = std::max(rows, m->md_rows) * 2 + 1024, p2 = std::max((size_t)npoints, m->md_points) * 2 + 256;
    int rc;
    if ((rc = m_alloc(&m->d_md, r2 * 32)) || (rc = m_alloc(&m->d_moff, p2 + 1)) || (rc = m_alloc(&m->d_mres, 2 * p2))) return rc;
    m->md_rows = r2, m->md_points = p2;
  }
  if (rows) UVO_HIP_CHECK(hipMemcpyAsync(m->d_md, desc, rows * 32, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_moff, offsets, (size_t)(npoints + 1) * 4, hipMemcpyHostToDevice, s));
  launch_medoid(s, m->d_md, m->d_moff, npoints, m->d_mres, m->d_mres + npoints);
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(best_idx, m->d_mres, (size_t)npoints * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(best_median, m->d_mres + npoints, (size_t)npoints * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  return UVO_OK;
}

int uvo_search_by_projection(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int min_x, int min_y, int max_x, int max_y,
                             int32_t* assigned, int nmp, const float* proj_x, const float* proj_y, const int32_t* level,
                             const float* view_cos, const uint8_t* in_view, const uint8_t* mp_desc, const float* scale_factors,
                             int nlevels, float th, float nnratio, int* n_matches) {
  if (!m || !n_matches) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_matches = 0;
  if (n < 0 || nmp < 0 || n > m->cfg.max_query || nmp > m->cfg.max_map_points || nlevels < 1 || nlevels > kMaxLevels * 4 || max_x <= min_x ||
      max_y <= min_y)
    return matcher_fail(UVO_E_BADARG, "sizes outside handle capacity");
  if (n == 0 || nmp == 0) return UVO_OK;
  if (!kp || !desc || !assigned || !proj_x || !proj_y || !level || !view_cos || !in_view || !mp_desc || !scale_factors)
    return matcher_fail(UVO_E_BADARG, "null pointer");
  for (int i = 0; i < nmp; ++i)
    if (in_view[i] && (level[i] < 0 || level[i] >= nlevels)) return matcher_fail(UVO_E_BADARG, "map point level outside 0..nlevels-1");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_kp, kp, sizeof(uvo_keypoint) * n, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_q, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_assigned, assigned, (size_t)n * 4, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_px, proj_x, (size_t)nmp * 4, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_py, proj_y, (size_t)nmp * 4, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_vc, view_cos, (size_t)nmp * 4, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_level, level, (size_t)nmp * 4, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_inview, in_view, (size_t)nmp, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_mpdesc, mp_desc, (size_t)nmp * 32, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemcpyAsync(m->d_scale, scale_factors, (size_t)nlevels * 4, hipMemcpyHostToDevice, s));
  auto go = [&](int stage) {
    launch_sbp(s, m->d_kp, n, m->d_q, min_x, min_y, max_x, max_y, m->d_assigned, nmp, m->d_px, m->d_py, m->d_level, m->d_vc, m->d_inview,
               m->d_mpdesc, m->d_scale, th, nnratio, m->d_cell_start, m->d_cell_items, m->d_cell_of_kp, m->d_cand_cnt, m->d_cand_start,
               m->d_cand, m->d_owner, m->d_owner_next, m->d_choice, m->d_nm, stage, (int64_t)m->cand_elems);
  };
  go(0);
  UVO_HIP_CHECK(hipGetLastError());
  int32_t total = 0;
  UVO_HIP_CHECK(hipMemcpyAsync(&total, m->d_cand_start + nmp, 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  if ((size_t)total > m->cand_elems) {
    if (m->d_cand) hipFree(m->d_cand);
    m->d_cand = nullptr;
    const size_t want = (size_t)total + total / 2 + 1024;
    int rc = m_alloc(&m->d_cand, want);
    if (rc) return rc;
    m->cand_elems = want;
  }
  go(1);
  UVO_HIP_CHECK(hipGetLastError());
  int32_t nm = 0;
  UVO_HIP_CHECK(hipMemcpyAsync(assigned, m->d_assigned, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipMemcpyAsync(&nm, m->d_nm, 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  *n_matches = nm;
  return UVO_OK;
}

// Tracking::SearchReferencePointsInFrustum (src/Tracking.cc:2176-2230) as one call: FrameKTL::isInFrustum on every local map
// point, then SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th) on the ones in view.  All inputs travel as one packed block
// through a pinned mirror, the projection results never leave the device, and the host waits once.
int uvo_search_points_in_frustum(uvo_matcher* m, const uvo_keypoint* kp, int n, const uint8_t* desc, int32_t* assigned, const uvo_camera_pose* cam,
                                 int npts, const float* xyz, const float* normal, const float* min_distance_inv, const float* max_distance_inv,
                                 const float* max_distance, const uint8_t* usable, const uint8_t* mp_desc, const float* scale_factors, int nlevels,
                                 float scale_factor, float viewing_cos_limit, float th, float nnratio, uint8_t* in_view, float* proj_x,
                                 float* proj_y, int32_t* level, float* view_cos, int* n_to_match, int* n_matches) {
  if (!m || !n_matches || !cam) return matcher_fail(UVO_E_BADARG, "null pointer");
  *n_matches = 0;
  if (n_to_match) *n_to_match = 0;
  const int min_x = (int)cam->min_x, min_y = (int)cam->min_y, max_x = (int)cam->max_x, max_y = (int)cam->max_y;
  if (n < 0 || npts < 0 || n > m->cfg.max_query || npts > m->cfg.max_map_points || nlevels < 1 || nlevels > kMaxLevels * 4 || max_x <= min_x ||
      max_y <= min_y)
    return matcher_fail(UVO_E_BADARG, "sizes outside handle capacity");
  if (npts == 0) return UVO_OK;
  if (!xyz || !normal || !min_distance_inv || !max_distance_inv || !max_distance || !mp_desc || !scale_factors)
    return matcher_fail(UVO_E_BADARG, "null pointer");
  if (n > 0 && (!kp || !desc || !assigned)) return matcher_fail(UVO_E_BADARG, "null pointer");
  if (!(scale_factor > 1.0f)) return matcher_fail(UVO_E_BADARG, "scale_factor must be > 1");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // ---- packed layout (offsets in bytes, every array 16-byte aligned) ----
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 15) & ~(size_t)15;
    return o;
  };
  const size_t N = (size_t)n, P = (size_t)npts;
  const size_t o_kp = take(N * sizeof(uvo_keypoint)), o_desc = take(N * 32), o_asg = take(N * 4), o_xyz = take(P * 12), o_nrm = take(P * 12);
  const size_t o_min = take(P * 4), o_max = take(P * 4), o_raw = take(P * 4), o_use = take(P), o_mpd = take(P * 32), o_sf = take((size_t)nlevels * 4);
  const size_t in_bytes = off;
  const size_t o_valid = take(P), o_u = take(P * 4), o_v = take(P * 4), o_lvl = take(P * 4), o_vc = take(P * 4), o_tail = take(16);
  const size_t total_bytes = off;
  if (total_bytes > m->arena_bytes) {
    UVO_HIP_CHECK(hipStreamSynchronize(s));
    if (m->d_arena) hipFree(m->d_arena);
    if (m->h_arena) (void)hipHostFree(m->h_arena);
    m->d_arena = nullptr, m->h_arena = nullptr, m->arena_bytes = 0;
    const size_t want = total_bytes + total_bytes / 4;
    int rc = m_alloc(&m->d_arena, want);
    if (rc) return rc;
    void* hp = nullptr;
    if (hipHostMalloc(&hp, want, hipHostMallocDefault) != hipSuccess) return matcher_fail(UVO_E_NOMEM, "pinned staging allocation failed");
    m->h_arena = (uint8_t*)hp, m->arena_bytes = want;
  }
  uint8_t *H = m->h_arena, *D = m->d_arena;
  if (n > 0) {
    std::memcpy(H + o_kp, kp, N * sizeof(uvo_keypoint));
    std::memcpy(H + o_desc, desc, N * 32);
    std::memcpy(H + o_asg, assigned, N * 4);
  }
  std::memcpy(H + o_xyz, xyz, P * 12);
  std::memcpy(H + o_nrm, normal, P * 12);
  std::memcpy(H + o_min, min_distance_inv, P * 4);
  std::memcpy(H + o_max, max_distance_inv, P * 4);
  std::memcpy(H + o_raw, max_distance, P * 4);
  if (usable)
    std::memcpy(H + o_use, usable, P);
  else
    std::memset(H + o_use, 1, P);
  std::memcpy(H + o_mpd, mp_desc, P * 32);
  std::memcpy(H + o_sf, scale_factors, (size_t)nlevels * 4);
  UVO_HIP_CHECK(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, s));
  uint8_t* d_valid = D + o_valid;
  float *d_u = (float*)(D + o_u), *d_v = (float*)(D + o_v), *d_vc = (float*)(D + o_vc);
  int32_t *d_lvl = (int32_t*)(D + o_lvl), *d_asg = (int32_t*)(D + o_asg);
  const float* d_sf = (const float*)(D + o_sf);
  {
    Profiler::Scope ps(&m->prof, "k_project", s);
    launch_project(s, UVO_PROJECT_FRUSTUM, *cam, npts, (const float*)(D + o_xyz), (const float*)(D + o_nrm), (const float*)(D + o_min),
                   (const float*)(D + o_max), (const float*)(D + o_raw), D + o_use, d_sf, nlevels, uvo_logf(scale_factor), viewing_cos_limit, d_valid,
                   d_u, d_v, d_lvl, d_vc);
  }
  int32_t total = 0, nm = 0;
  UVO_HIP_CHECK(hipGetLastError());
  if (n > 0) {
    // candidate lists: sized for 8 per map point up front; a denser frame is detected from the returned total and the match
    // stage is repeated once with a larger buffer (the kernels never write or read past the capacity they are given)
    if (m->cand_elems < P * 8) {
      UVO_HIP_CHECK(hipStreamSynchronize(s));
      if (m->d_cand) hipFree(m->d_cand);
      m->d_cand = nullptr, m->cand_elems = 0;
      int rc = m_alloc(&m->d_cand, P * 8);
      if (rc) return rc;
      m->cand_elems = P * 8;
    }
    auto go = [&](int stage) {
      Profiler::Scope ps(&m->prof, stage ? "k_sbp_match" : "k_sbp_count", s);
      launch_sbp(s, (const uvo_keypoint*)(D + o_kp), n, D + o_desc, min_x, min_y, max_x, max_y, d_asg, npts, d_u, d_v, d_lvl, d_vc, d_valid, D + o_mpd,
                 d_sf, th, nnratio, m->d_cell_start, m->d_cell_items, m->d_cell_of_kp, m->d_cand_cnt, m->d_cand_start, m->d_cand, m->d_owner,
                 m->d_owner_next, m->d_choice, m->d_nm, stage, (int64_t)m->cand_elems);
    };
    auto fetch = [&]() -> int {
      UVO_HIP_CHECK(hipGetLastError());
      UVO_HIP_CHECK(hipMemcpyAsync(H + o_asg, d_asg, N * 4, hipMemcpyDeviceToHost, s));
      UVO_HIP_CHECK(hipMemcpyAsync(H + o_tail, m->d_cand_start + npts, 4, hipMemcpyDeviceToHost, s));
      UVO_HIP_CHECK(hipMemcpyAsync(H + o_tail + 4, m->d_nm, 4, hipMemcpyDeviceToHost, s));
      UVO_HIP_CHECK(hipMemcpyAsync(H + o_valid, d_valid, o_tail - o_valid, hipMemcpyDeviceToHost, s));
      UVO_HIP_CHECK(hipStreamSynchronize(s));
      return UVO_OK;
    };
    go(0);
    go(1);
    int rc = fetch();
    if (rc) return rc;
    std::memcpy(&total, H + o_tail, 4);
    if ((size_t)total > m->cand_elems) {
      hipFree(m->d_cand);
      m->d_cand = nullptr, m->cand_elems = 0;
      const size_t want = (size_t)total + total / 2 + 1024;
      rc = m_alloc(&m->d_cand, want);
      if (rc) return rc;
      m->cand_elems = want;
      UVO_HIP_CHECK(hipMemcpyAsync(d_asg, assigned, N * 4, hipMemcpyHostToDevice, s));  // undo the truncated run's assignments
      go(1);
      rc = fetch();
      if (rc) return rc;
    }
    std::memcpy(&nm, H + o_tail + 4, 4);
    std::memcpy(assigned, H + o_asg, N * 4);
  } else {
    UVO_HIP_CHECK(hipMemcpyAsync(H + o_valid, d_valid, o_tail - o_valid, hipMemcpyDeviceToHost, s));
    UVO_HIP_CHECK(hipStreamSynchronize(s));
  }
  int to_match = 0;
  for (size_t i = 0; i < P; ++i) to_match += H[o_valid + i] != 0;
  if (n_to_match) *n_to_match = to_match;
  if (in_view) std::memcpy(in_view, H + o_valid, P);
  if (proj_x) std::memcpy(proj_x, H + o_u, P * 4);
  if (proj_y) std::memcpy(proj_y, H + o_v, P * 4);
  if (level) std::memcpy(level, H + o_lvl, P * 4);
  if (view_cos) std::memcpy(view_cos, H + o_vc, P * 4);
  *n_matches = nm;
  return UVO_OK;
}

hipStream_t uvo_matcher_stream_internal(uvo_matcher* m) { return m->stream; }
// Called by the extractor a matcher is attached to: whenever a batch moves to another pipeline lane the followers move with it (a
// matcher that kept the previous lane's stream would read the new batch's descriptors with no ordering at all), and when the
// extractor is destroyed they go back to their own streams instead of keeping a dead one.
void uvo_matcher_follow_internal(uvo_matcher* m, hipStream_t s) {
  if (m->stream == s) return;
  if (m->prof.on) (void)hipStreamSynchronize(m->stream);  // the profiler's open events belong to the stream they were recorded on
  m->stream = s;
}
void uvo_matcher_orphaned_internal(uvo_matcher* m) {
  m->attached_to = nullptr;
  m->stream = m->own_stream;
}

int uvo_matcher_wait_extractor(uvo_matcher* m, uvo_extractor* h) {
  if (!m || !h) return matcher_fail(UVO_E_BADARG, "null handle");
  if (uvo_extractor_device_internal(h) != m->device) return matcher_fail(UVO_E_BADARG, "handles live on different devices");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  UVO_HIP_CHECK(hipEventRecord(m->ev, uvo_extractor_stream_internal(h)));
  UVO_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev, 0));
  return UVO_OK;
}

int uvo_extractor_wait_matcher(uvo_extractor* h, uvo_matcher* m) {
  if (!m || !h) return matcher_fail(UVO_E_BADARG, "null handle");
  if (uvo_extractor_device_internal(h) != m->device) return matcher_fail(UVO_E_BADARG, "handles live on different devices");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  UVO_HIP_CHECK(hipEventRecord(m->ev, m->stream));
  UVO_HIP_CHECK(hipStreamWaitEvent(uvo_extractor_stream_internal(h), m->ev, 0));
  return UVO_OK;
}

// Event hand-offs between two queues cost tens of microseconds each on this runtime (tools/step_trace_summary.py: 0.32 ms between a
// lane's k_describe and its next k_pad_level0 with the matcher on its own stream -- two hand-offs around a 0.13 ms kernel); in the
// extractor lane's own stream the matcher's kernels simply queue up behind the batch that feeds them.
int uvo_matcher_attach_extractor(uvo_matcher* m, uvo_extractor* h) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  if (h && uvo_extractor_device_internal(h) != m->device) return matcher_fail(UVO_E_BADARG, "handles live on different devices");
  if (m->prof.on) UVO_HIP_CHECK(hipStreamSynchronize(m->stream));  // the profiler's open events belong to the stream they were recorded on
  if (m->attached_to && m->attached_to != h) uvo_extractor_drop_follower_internal(m->attached_to, m);
  m->attached_to = h;
  if (!h) {
    m->stream = m->own_stream;
    return UVO_OK;
  }
  uvo_extractor_add_follower_internal(h, m);  // from now on the extractor moves this handle's stream with its batches
  m->stream = uvo_extractor_stream_internal(h);
  return UVO_OK;
}

int uvo_matcher_profile(uvo_matcher* m, int enable) {
  if (!m) return matcher_fail(UVO_E_BADARG, "null handle");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  UVO_HIP_CHECK(hipStreamSynchronize(m->stream));
  m->prof.on = enable != 0;
  m->prof.clear();
  return UVO_OK;
}

int uvo_matcher_kernel_times(uvo_matcher* m, char* names, int names_cap, float* ms, int32_t* launches, int cap, int* n) {
  if (!m || !names || !ms || !launches || !n) return matcher_fail(UVO_E_BADARG, "null pointer");
  UVO_HIP_CHECK(hipSetDevice(m->device));
  UVO_HIP_CHECK(hipStreamSynchronize(m->stream));
  *n = m->prof.report(names, names_cap, ms, launches, cap);
  return UVO_OK;
}

}  // extern "C"
