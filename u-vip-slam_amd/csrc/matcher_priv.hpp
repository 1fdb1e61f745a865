// Private to the matcher translation units (matcher.cpp, matcher_search.cpp): the handle behind uvo_matcher*.
#pragma once
#include <vector>

#include "common.hpp"
#include "profiler.hpp"

// growable device buffer owned by the handle (staging of variable-size inputs / candidate lists)
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct uvo_matcher {
  uvo_matcher_cfg cfg;
  int device = 0;
  hipStream_t stream = nullptr;      // where the handle's work is enqueued: its own stream, or an extractor lane's (uvo_matcher_attach_extractor)
  hipStream_t own_stream = nullptr;
  uvo_extractor* attached_to = nullptr;  // the extractor whose current lane `stream` follows (it keeps a list of its followers and lets go of them when it dies)
  // knn2 staging
  uint8_t *d_q = nullptr, *d_t = nullptr, *d_mask = nullptr;
  size_t mask_bytes = 0;
  int32_t *d_idx0 = nullptr, *d_idx1 = nullptr;
  uint16_t *d_d0 = nullptr, *d_d1 = nullptr, *d_dist = nullptr;
  size_t dist_elems = 0;
  // search-by-projection
  uvo_keypoint* d_kp = nullptr;
  float *d_px = nullptr, *d_py = nullptr, *d_vc = nullptr, *d_scale = nullptr;
  int32_t *d_level = nullptr, *d_assigned = nullptr, *d_cell_start = nullptr, *d_cell_items = nullptr, *d_cell_of_kp = nullptr;
  int32_t *d_cand_cnt = nullptr, *d_cand_start = nullptr, *d_owner = nullptr, *d_owner_next = nullptr, *d_choice = nullptr, *d_nm = nullptr;
  uint8_t *d_inview = nullptr, *d_mpdesc = nullptr;
  uint32_t* d_cand = nullptr;
  size_t cand_elems = 0;
  uint8_t* d_md = nullptr;  // medoid staging: descriptors, offsets, results
  int32_t *d_moff = nullptr, *d_mres = nullptr;
  size_t md_rows = 0, md_points = 0;
  hipEvent_t ev = nullptr;
  DevBuf scratch[32];  // staging: slots 0..23 matcher_search.cpp, 24..31 matcher_batch.cpp (see the slot enums there)
  void* tri_batch = nullptr;  // candidate lists of uvo_search_for_triangulation_batch (uvo::TriBatch, matcher_batch.cpp)
  // uvo_search_points_in_frustum: one packed input block (pinned host mirror -> device arena, one copy each way)
  uint8_t *d_arena = nullptr, *h_arena = nullptr;
  size_t arena_bytes = 0;
  uvo::Profiler prof;
};


namespace uvo {
int matcher_fail(int code, const char* msg);
void tri_batch_free(void* p);
template <class T>
static int m_alloc(T** p, size_t n) {
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) {
    hip_err_set(e, "hipMalloc");
    return e == hipErrorOutOfMemory ? UVO_E_NOMEM : UVO_E_HIP;
  }
  return UVO_OK;
}
}  // namespace uvo
