// Grid-bucketed FAST: the GPU form of Grider_FAST::perform_griding (include/Grider_FAST.h:81-137; OpenVINS helper, the
// call in src/Tracking.cc:940 is commented out, so this is the alternative bucketing mode, not the live path).
//   size = img / grid; every in-bounds ROI (x, y, size_x, size_y): cv::FAST(roi, threshold, nms); sort by response
//   descending; keep the first num_features/(grid_x*grid_y) + 1; shift by the ROI origin; concatenate ROI-major.
// The reference's std::sort (:117) is unstable, so equal responses come out in unspecified order; declared order here:
// response descending, then y, then x ascending (ROI coordinates).
//   k_grid_score : one thread per pixel, full 9/16 segment test + cornerScore for pixels in a ROI's 3-px-inset interior
//   k_grid_select: one workgroup per ROI -- optional 3x3 NMS inside the ROI, top-K by repeated workgroup arg-max
//                  (K is small: num_features / cells + 1), keypoints written at the ROI's prefix offset.
#include "common.hpp"

namespace uvo {

__device__ __forceinline__ int arc9_maxmin_g(const int* d) {
  int best = -(1 << 30);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    int m = 1 << 30;
#pragma unroll
    for (int k = 0; k < 9; ++k) m = min(m, d[(s + k) & 15]);
    best = max(best, m);
  }
  return best;
}

struct GridGeom {
  int w, h, size_x, size_y, ct_cols, ct_rows;
};

__global__ __launch_bounds__(256) void k_grid_score(const uint8_t* __restrict__ img, int64_t stride, GridGeom G, int threshold,
                                                    uint8_t* __restrict__ score) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= G.w || y >= G.h) return;
  int s = 0;
  const int rx = x / G.size_x, ry = y / G.size_y;
  const int lx = x - rx * G.size_x, ly = y - ry * G.size_y;
  // ROIs that stick out of the image are skipped (:104-105); ct_cols = floor(w / size_x) columns exist at all (:93-94)
  if (rx < G.ct_cols && ry < G.ct_rows && lx >= 3 && lx < G.size_x - 3 && ly >= 3 && ly < G.size_y - 3) {
    const uint8_t* p = img + (int64_t)y * stride + x;
    const int v = p[0];
    int d[16];
    d[0] = p[3 * stride], d[1] = p[3 * stride + 1], d[2] = p[2 * stride + 2], d[3] = p[stride + 3];
    d[4] = p[3], d[5] = p[-stride + 3], d[6] = p[-2 * stride + 2], d[7] = p[-3 * stride + 1];
    d[8] = p[-3 * stride], d[9] = p[-3 * stride - 1], d[10] = p[-2 * stride - 2], d[11] = p[-stride - 3];
    d[12] = p[-3], d[13] = p[stride - 3], d[14] = p[2 * stride - 2], d[15] = p[3 * stride - 1];
    int nd[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      d[k] -= v;
      nd[k] = -d[k];
    }
    const int best = max(arc9_maxmin_g(d), arc9_maxmin_g(nd));  // corner at t  <=>  best > t;  cornerScore = best - 1
    if (best > threshold) s = best;                             // stored as score + 1 so that 0 means "not a corner"
  }
  score[(int64_t)y * G.w + x] = (uint8_t)s;
}

// key: response desc, then y asc, then x asc (ROI coordinates, < 4096)
__device__ __forceinline__ uint32_t grid_key(int s, int ly, int lx) { return ((uint32_t)s << 24) | ((uint32_t)(4095 - ly) << 12) | (uint32_t)(4095 - lx); }

__global__ __launch_bounds__(256) void k_grid_select(const uint8_t* __restrict__ score, GridGeom G, int nms, int keep_per_cell,
                                                     uint32_t* __restrict__ lists, int32_t* __restrict__ counts, int pass,
                                                     uvo_keypoint* __restrict__ out, int cap, int32_t* __restrict__ n_out) {
  __shared__ uint32_t s_red[256];
  __shared__ int s_n;
  const int r = blockIdx.x;
  const int rx = r % G.ct_cols, ry = r / G.ct_cols;
  const int x0 = rx * G.size_x, y0 = ry * G.size_y;
  uint32_t* list = lists + (int64_t)r * G.size_x * G.size_y;
  const int tid = threadIdx.x;
  if (pass == 0) {
    // collect the ROI's corners (after NMS) into its list
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int iw = G.size_x - 6, ih = G.size_y - 6;
    for (int i = tid; i < iw * ih; i += 256) {
      const int lx = 3 + i % iw, ly = 3 + i / iw;
      const uint8_t* q = score + (int64_t)(y0 + ly) * G.w + x0 + lx;
      if (q[0] == 0) continue;
      const int s = q[0] - 1;  // plane holds score + 1
      if (nms) {
        // strict maximum over the 8 neighbours; non-corners (and everything outside the ROI interior, which holds 0 in the
        // plane because other ROIs' interiors are >= 6 px away) count as score 0
        auto sc = [](int v) { return v > 0 ? v - 1 : 0; };
        const bool keep = s > sc(q[-G.w - 1]) && s > sc(q[-G.w]) && s > sc(q[-G.w + 1]) && s > sc(q[-1]) && s > sc(q[1]) &&
                          s > sc(q[G.w - 1]) && s > sc(q[G.w]) && s > sc(q[G.w + 1]);
        if (!keep) continue;
      }
      list[atomicAdd(&s_n, 1)] = grid_key(s, ly, lx);
    }
    __syncthreads();
    if (tid == 0) counts[r] = s_n;
    return;
  }
  // pass 1: top-K of the list by repeated arg-max, written at the ROI-major prefix offset
  const int n = counts[r];
  const int k_take = n < keep_per_cell ? n : keep_per_cell;
  int offset = 0;
  for (int i = 0; i < r; ++i) offset += counts[i] < keep_per_cell ? counts[i] : keep_per_cell;
  if (r == G.ct_cols * G.ct_rows - 1 && tid == 0) *n_out = offset + k_take;
  uint32_t last = 0xFFFFFFFFu;  // keys are distinct: take the largest key below the previous one
  for (int k = 0; k < k_take; ++k) {
    uint32_t best = 0;
    for (int i = tid; i < n; i += 256) {
      const uint32_t key = list[i];
      if (key < last && key > best) best = key;
    }
    s_red[tid] = best;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) s_red[tid] = max(s_red[tid], s_red[tid + st]);
      __syncthreads();
    }
    last = s_red[0];
    __syncthreads();
    if (tid == 0 && offset + k < cap) {
      uvo_keypoint kp;
      kp.x = (float)(x0 + (4095 - (int)(last & 0xfff)));
      kp.y = (float)(y0 + (4095 - (int)((last >> 12) & 0xfff)));
      kp.size = 7.f, kp.angle = -1.f, kp.response = (float)(last >> 24), kp.octave = 0, kp.class_id = -1;
      out[offset + k] = kp;
    }
  }
}

void launch_grider(hipStream_t s, const uint8_t* d_img, int w, int h, int64_t stride, int num_features, int grid_x, int grid_y, int threshold,
                   int nms, uint8_t* d_score, uint32_t* d_lists, int32_t* d_counts, uvo_keypoint* d_out, int cap, int32_t* d_n_out) {
  GridGeom G;
  G.w = w, G.h = h, G.size_x = w / grid_x, G.size_y = h / grid_y;
  G.ct_cols = w / G.size_x, G.ct_rows = h / G.size_y;
  const int keep = num_features / (grid_x * grid_y) + 1;
  threshold = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
  hipLaunchKernelGGL(k_grid_score, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, d_img, stride, G, threshold, d_score);
  const int rois = G.ct_cols * G.ct_rows;
  hipLaunchKernelGGL(k_grid_select, dim3(rois), dim3(256), 0, s, d_score, G, nms, keep, d_lists, d_counts, 0, d_out, cap, d_n_out);
  hipLaunchKernelGGL(k_grid_select, dim3(rois), dim3(256), 0, s, d_score, G, nms, keep, d_lists, d_counts, 1, d_out, cap, d_n_out);
}

}  // namespace uvo
