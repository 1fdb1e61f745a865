// ORBmatcher::SearchByProjection(FrameKTL&, const vector<MapPoint*>&, th) on the GPU
// (src/ORBmatcher.cc:49-125; frame grid src/FrameKTL.cc:83-84,250-264,359-436).
//
//   k_grid_build : 64x48 cell lists of the frame keypoints; PosInGrid uses round() (:428-429); list order inside a
//                  cell is keypoint index order (push_back order, :258-263)
//   k_sbp_count / k_sbp_fill : per map point, the candidate list GetFeaturesInArea would return, in its
//                  (ix, iy, insertion) order, with the 256-bit Hamming distance of every candidate (CSR)
//   k_sbp_resolve: the reference's greedy loop is order dependent (`if(F.mvpMapPoints[idx]) continue;` :91 and the
//                  write at :119).  It is solved exactly as a fixed point: owner[k] = lowest-index map point whose
//                  accepted choice is keypoint k; map point i may not use k when owner[k] < i.  Map point 0 is final
//                  after one sweep, map point i after at most i+1, so the iteration ends in the sequential result;
//                  in practice dependency chains are 2-4 deep.
#include "common.hpp"

namespace uvo {

constexpr int GR_COLS = 64, GR_ROWS = 48;  // include/FrameKTL.h:45-46
constexpr int TH_HIGH = 100;               // src/ORBmatcher.cc:40

struct SbpFrame {
  const uvo_keypoint* kp;
  const uint8_t* desc;
  int n;
  int min_x, min_y;
  float inv_w, inv_h;
};

// One workgroup of 1024 threads.  Frames of up to GB_ITEMS key points keep the item list and the cell of every key point in LDS,
// so the scatter and the per-cell insertion sorts never wait on HBM; larger frames run the same phases on the output arrays.
constexpr int GB_THREADS = 1024, GB_ITEMS = 4096;
__global__ __launch_bounds__(GB_THREADS) void k_grid_build(SbpFrame F, int32_t* __restrict__ cell_start, int32_t* __restrict__ cell_items,
                                                           int32_t* __restrict__ cell_of_kp) {
  constexpr int NC = GR_COLS * GR_ROWS, PER = NC / GB_THREADS;  // 3072 cells, 3 per thread
  static_assert(NC % GB_THREADS == 0, "cells must divide evenly over the threads");
  __shared__ int s_cnt[NC];
  __shared__ int s_cur[NC];
  __shared__ int s_wave[GB_THREADS / 64];
  __shared__ int32_t s_items[GB_ITEMS];
  __shared__ int16_t s_cell[GB_ITEMS];
  const int tid = threadIdx.x;
  const bool in_lds = F.n <= GB_ITEMS;
  // phase 1: cell of every keypoint, cell populations
  for (int c = tid; c < NC; c += GB_THREADS) s_cnt[c] = 0, s_cur[c] = 0;
  __syncthreads();
  for (int i = tid; i < F.n; i += GB_THREADS) {
    const int px = (int)roundf((F.kp[i].x - (float)F.min_x) * F.inv_w);
    const int py = (int)roundf((F.kp[i].y - (float)F.min_y) * F.inv_h);
    int c = -1;
    if (px >= 0 && px < GR_COLS && py >= 0 && py < GR_ROWS) {
      c = px * GR_ROWS + py;
      atomicAdd(&s_cnt[c], 1);
    }
    cell_of_kp[i] = c;
    if (in_lds) s_cell[i] = (int16_t)c;
  }
  __syncthreads();
  // phase 2: exclusive scan of the 3072 populations (3 consecutive cells per thread, wavefront scan, 16 partials)
  int local = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) local += s_cnt[tid * PER + k];
  int incl = local;
  const int lane = tid & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_wave[tid >> 6] = incl;
  __syncthreads();
  int run = incl - local;
  for (int q = 0; q < (tid >> 6); ++q) run += s_wave[q];
  int first[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    first[k] = run;
    cell_start[tid * PER + k] = run;
    run += s_cnt[tid * PER + k];
  }
  if (tid == GB_THREADS - 1) cell_start[NC] = run;
  // s_cur doubles as the cells' first slots for the scatter of other threads
#pragma unroll
  for (int k = 0; k < PER; ++k) s_cur[tid * PER + k] = first[k];
  __syncthreads();
  // phase 3: scatter (arbitrary order inside a cell), phase 4: every cell's short list sorted ascending = push_back order
  int32_t* items = in_lds ? s_items : cell_items;
  for (int i = tid; i < F.n; i += GB_THREADS) {
    const int c = in_lds ? (int)s_cell[i] : cell_of_kp[i];
    if (c >= 0) items[atomicAdd(&s_cur[c], 1)] = i;
  }
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < PER; ++k) {
    const int n = s_cnt[tid * PER + k];
    if (n < 2) continue;
    int32_t* a = items + first[k];
    for (int i = 1; i < n; ++i) {
      const int v = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > v) {
        a[j + 1] = a[j];
        --j;
      }
      a[j + 1] = v;
    }
  }
  if (in_lds) {
    __syncthreads();
    for (int i = tid; i < F.n; i += GB_THREADS) cell_items[i] = s_items[i];  // key points outside the grid leave the tail unused
  }
}

struct SbpMap {
  const float* proj_x;
  const float* proj_y;
  const int32_t* level;
  const float* view_cos;
  const uint8_t* in_view;
  const uint8_t* desc;
  int n;
};

// window of GetFeaturesInArea (:364-382); returns false when the query leaves the grid
__device__ __forceinline__ bool sbp_window(const SbpFrame& F, float x, float y, float r, int& x0, int& x1, int& y0, int& y1) {
  x0 = (int)floorf((x - (float)F.min_x - r) * F.inv_w);
  x0 = max(0, x0);
  if (x0 >= GR_COLS) return false;
  x1 = (int)ceilf((x - (float)F.min_x + r) * F.inv_w);
  x1 = min(GR_COLS - 1, x1);
  if (x1 < 0) return false;
  y0 = (int)floorf((y - (float)F.min_y - r) * F.inv_h);
  y0 = max(0, y0);
  if (y0 >= GR_ROWS) return false;
  y1 = (int)ceilf((y - (float)F.min_y + r) * F.inv_h);
  y1 = min(GR_ROWS - 1, y1);
  if (y1 < 0) return false;
  return true;
}

__device__ __forceinline__ int hamming256(const uint8_t* a, const uint8_t* b) {
  const uint4* A = reinterpret_cast<const uint4*>(a);
  const uint4* B = reinterpret_cast<const uint4*>(b);
  const uint4 a0 = A[0], a1 = A[1], b0 = B[0], b1 = B[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
         __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// FILL = false: count candidates per map point; FILL = true: write packed candidates at cand_start[i]
// packed candidate: keypoint index (16 bits) | distance (9 bits) << 16 | octave (6 bits) << 25
template <bool FILL>
__global__ __launch_bounds__(256) void k_sbp_cand(SbpFrame F, SbpMap M, const int32_t* __restrict__ cell_start,
                                                  const int32_t* __restrict__ cell_items, const float* __restrict__ scale_factors, float th,
                                                  int32_t* __restrict__ cand_cnt, const int32_t* __restrict__ cand_start,
                                                  uint32_t* __restrict__ cand, int64_t cand_cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M.n) return;
  int n = 0;
  if (M.in_view[i]) {
    const int lvl = M.level[i];
    float r = ((double)M.view_cos[i] > 0.998) ? 2.5f : 4.0f;  // RadiusByViewingCos :127-133
    if (th != 1.0f) r *= th;
    r = r * scale_factors[lvl];
    const float x = M.proj_x[i], y = M.proj_y[i];
    const int minLevel = lvl - 1, maxLevel = lvl;
    int x0, x1, y0, y1;
    if (sbp_window(F, x, y, r, x0, x1, y0, y1)) {
      const int o = FILL ? cand_start[i] : 0;
      uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
      if (FILL) {
        const uint4* Q = reinterpret_cast<const uint4*>(M.desc + (int64_t)i * 32);
        q0 = Q[0], q1 = Q[1];
      }
      // the cells (ix, y0..y1) of one grid column are consecutive in the CSR arrays, so a column is one contiguous run of items in
      // GetFeaturesInArea's own (ix, iy, insertion) order; the run is walked eight items at a time with the index, key point and
      // descriptor loads of a batch issued together
      for (int ix = x0; ix <= x1; ++ix) {
        const int k_end = cell_start[ix * GR_ROWS + y1 + 1];
        for (int k0 = cell_start[ix * GR_ROWS + y0]; k0 < k_end; k0 += 8) {
          int idx[8], oct[8];
          float kx[8], ky[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) idx[u] = k0 + u < k_end ? cell_items[k0 + u] : -1;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uvo_keypoint* p = F.kp + (idx[u] >= 0 ? idx[u] : 0);
            kx[u] = p->x, ky[u] = p->y, oct[u] = p->octave;
          }
          bool take[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            take[u] = idx[u] >= 0 && !(oct[u] < minLevel || oct[u] > maxLevel) && !(fabsf(kx[u] - x) > r || fabsf(ky[u] - y) > r);
          if (FILL) {
            uint4 d0[8], d1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const uint4* D = reinterpret_cast<const uint4*>(F.desc + (int64_t)(take[u] ? idx[u] : 0) * 32);
              d0[u] = D[0], d1[u] = D[1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (!take[u]) continue;
              const int d = __popc(q0.x ^ d0[u].x) + __popc(q0.y ^ d0[u].y) + __popc(q0.z ^ d0[u].z) + __popc(q0.w ^ d0[u].w) +
                            __popc(q1.x ^ d1[u].x) + __popc(q1.y ^ d1[u].y) + __popc(q1.z ^ d1[u].z) + __popc(q1.w ^ d1[u].w);
              // a list that outgrows the buffer is dropped here and the host repeats the stage with a larger one
              if ((int64_t)o + n < cand_cap) cand[o + n] = (uint32_t)idx[u] | ((uint32_t)d << 16) | ((uint32_t)(oct[u] & 63) << 25);
              ++n;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) n += take[u] ? 1 : 0;
          }
        }
      }
    }
  }
  if (!FILL) cand_cnt[i] = n;
}

__global__ __launch_bounds__(1024) void k_scan_i32(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n) {
  // single-workgroup exclusive scan, out[n] = total
  __shared__ int s_part[1024];
  const int per = (n + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(b + per, n);
  int s = 0;
  for (int i = b; i < e; ++i) s += in[i];
  s_part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
  for (int i = b; i < e; ++i) {
    const int v = in[i];
    out[i] = run;
    run += v;
  }
  if (threadIdx.x == 1023) out[n] = s_part[1023];
}

__global__ __launch_bounds__(1024) void k_sbp_resolve(int nkp, int nmp, const int32_t* __restrict__ cand_start, const uint32_t* __restrict__ cand,
                                                      int64_t cand_cap, float nnratio, int32_t* __restrict__ assigned, int32_t* owner,
                                                      int32_t* owner_next, int32_t* __restrict__ choice,
                                                      int32_t* __restrict__ n_matches) {
  __shared__ int s_changed, s_count;
  // ownership tables in LDS for frames of up to 4096 key points (the atomics and the dependent reads of the walk stay on chip)
  __shared__ int32_t s_owner[4096], s_owner_next[4096];
  if (nkp <= 4096) owner = s_owner, owner_next = s_owner_next;
  const int INF = 0x7fffffff;
  // owner: -1 = held before the call (F.mvpMapPoints[idx] already set), INF = free
  for (int k = threadIdx.x; k < nkp; k += blockDim.x) owner[k] = assigned[k] >= 0 ? -1 : INF;
  for (int i = threadIdx.x; i < nmp; i += blockDim.x) choice[i] = -2;
  __syncthreads();
  for (int iter = 0; iter <= nmp; ++iter) {
    if (threadIdx.x == 0) s_changed = 0;
    for (int k = threadIdx.x; k < nkp; k += blockDim.x) owner_next[k] = owner[k] < 0 ? -1 : INF;
    __syncthreads();
    for (int i = threadIdx.x; i < nmp; i += blockDim.x) {
      int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
      const int c_end = (int)min((int64_t)cand_start[i + 1], cand_cap);
      // eight candidates at a time: their words, then their owners, are fetched as independent loads before the (ordered) walk,
      // instead of two dependent memory round trips per candidate
      for (int c0 = cand_start[i]; c0 < c_end; c0 += 8) {
        uint32_t vv[8];
        int ow[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) vv[k] = c0 + k < c_end ? cand[c0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) ow[k] = c0 + k < c_end ? owner[vv[k] & 0xffffu] : -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (c0 + k >= c_end) break;
          const uint32_t v = vv[k];
          const int idx = (int)(v & 0xffffu);
          if (ow[k] < i) continue;  // taken before this map point's turn
          const int dist = (int)((v >> 16) & 0x1ffu), oct = (int)(v >> 25);
          if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestLevel2 = bestLevel;
            bestLevel = oct;
            bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = oct;
            bestDist2 = dist;
          }
        }
      }
      int ch = -1;
      if (bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)) ch = bestIdx;
      if (ch != choice[i]) {
        choice[i] = ch;
        s_changed = 1;
      }
      if (ch >= 0) atomicMin(&owner_next[ch], i);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nkp; k += blockDim.x) owner[k] = owner_next[k];
    const int changed = s_changed;
    __syncthreads();
    if (!changed) break;
  }
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < nkp; k += blockDim.x) {
    const int o = owner[k];
    if (o >= 0 && o != INF) {
      assigned[k] = o;
      atomicAdd(&s_count, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *n_matches = s_count;
}

void launch_grid_build(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y,
                       int32_t* d_cell_start, int32_t* d_cell_items, int32_t* d_cell_of_kp) {
  SbpFrame F{d_kp, d_desc, n, min_x, min_y, (float)GR_COLS / (float)(max_x - min_x), (float)GR_ROWS / (float)(max_y - min_y)};
  hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(GB_THREADS), 0, s, F, d_cell_start, d_cell_items, d_cell_of_kp);
}

void launch_sbp(hipStream_t s, const uvo_keypoint* d_kp, int n, const uint8_t* d_desc, int min_x, int min_y, int max_x, int max_y,
                int32_t* d_assigned, int nmp, const float* d_px, const float* d_py, const int32_t* d_level, const float* d_vc,
                const uint8_t* d_inview, const uint8_t* d_mpdesc, const float* d_scale, float th, float nnratio, int32_t* d_cell_start,
                int32_t* d_cell_items, int32_t* d_cell_of_kp, int32_t* d_cand_cnt, int32_t* d_cand_start, uint32_t* d_cand, int32_t* d_owner,
                int32_t* d_owner_next, int32_t* d_choice, int32_t* d_n_matches, int stage, int64_t cand_cap) {
  SbpFrame F{d_kp, d_desc, n, min_x, min_y, (float)GR_COLS / (float)(max_x - min_x), (float)GR_ROWS / (float)(max_y - min_y)};
  SbpMap M{d_px, d_py, d_level, d_vc, d_inview, d_mpdesc, nmp};
  const int blocks = (nmp + 255) / 256;
  if (stage == 0) {
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(GB_THREADS), 0, s, F, d_cell_start, d_cell_items, d_cell_of_kp);
    hipLaunchKernelGGL(k_sbp_cand<false>, dim3(blocks), dim3(256), 0, s, F, M, d_cell_start, d_cell_items, d_scale, th, d_cand_cnt,
                       (const int32_t*)nullptr, (uint32_t*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, s, d_cand_cnt, d_cand_start, nmp);
  } else {
    hipLaunchKernelGGL(k_sbp_cand<true>, dim3(blocks), dim3(256), 0, s, F, M, d_cell_start, d_cell_items, d_scale, th, d_cand_cnt, d_cand_start,
                       d_cand, cand_cap);
    hipLaunchKernelGGL(k_sbp_resolve, dim3(1), dim3(1024), 0, s, n, nmp, d_cand_start, d_cand, cand_cap, nnratio, d_assigned, d_owner,
                       d_owner_next, d_choice, d_n_matches);
  }
}

}  // namespace uvo
