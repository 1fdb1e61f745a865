// 256-bit Hamming matching kernels.
//   k_knn2   : all-pairs best / second-best per query -- cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) as used by
//              Utils::ratioMatching (include/utils.h:81-111); distance = ORBmatcher::DescriptorDistance
//              (src/ORBmatcher.cc:1794-1810: popcount of the XOR over 8 x 32 bits).
//   k_matrix : full distance matrix (MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:236-247).
// Two forms of the all-pairs distance:
//   * k_knn2_mfma (no mask): the all-pairs Hamming distance IS a dense contraction over the 256 bit positions --
//     d(q, t) = pop(q) + pop(t) - 2 * <q, t> with the descriptors taken as 0/1 vectors -- so the inner products run on the
//     matrix cores: v_mfma_i32_32x32x32_i8 on descriptors expanded to one byte per bit (the train tile expanded once per
//     workgroup into LDS, the query fragments once per wavefront into registers).  Exact in int32.
//   * k_knn2 / k_matrix / k_medoid: integer VALU (v_xor_b32 + v_bcnt_u32_b32); one lane owns one query descriptor in
//     8 VGPRs, train descriptors are staged through LDS in tiles of 128 and read back as wave-uniform broadcasts.
#include "common.hpp"

namespace uvo {

constexpr int HM_TILE = 128;  // train descriptors per LDS tile (4 KiB)

__device__ __forceinline__ int popc256(const uint4& qa, const uint4& qb, const uint4& ta, const uint4& tb) {
  int d = __popc(qa.x ^ ta.x);
  d += __popc(qa.y ^ ta.y);
  d += __popc(qa.z ^ ta.z);
  d += __popc(qa.w ^ ta.w);
  d += __popc(qb.x ^ tb.x);
  d += __popc(qb.y ^ tb.y);
  d += __popc(qb.z ^ tb.z);
  d += __popc(qb.w ^ tb.w);
  return d;
}

__device__ __forceinline__ int clamp_count(int n, int cap) { return n < 0 ? 0 : (n > cap ? cap : n); }

// grid: (ceil(max_query/256), pairs).  Ties keep the lower train index (strict <, ascending scan).
__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq_arr, int nq_fixed, int q_stride,
                                              const uint8_t* __restrict__ t, const int32_t* __restrict__ nt_arr, int nt_fixed, int t_stride,
                                              const uint8_t* __restrict__ mask, int out_stride, int32_t* __restrict__ idx0,
                                              uint16_t* __restrict__ d0, int32_t* __restrict__ idx1, uint16_t* __restrict__ d1) {
  __shared__ uint4 s_t[HM_TILE * 2];
  const int pair = blockIdx.y;
  // counts come from device arrays (an extractor's n_out may exceed the caller's capacity): never walk past a slice
  // (fixed counts were validated on the host)
  const int nq = nq_arr ? clamp_count(nq_arr[pair], q_stride < out_stride ? q_stride : out_stride) : nq_fixed;
  const int nt = nt_arr ? clamp_count(nt_arr[pair], t_stride < 65535 ? t_stride : 65535) : nt_fixed;  // the key packs the train index in 16 bits
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= nq) return;
  const uint4* Q = reinterpret_cast<const uint4*>(q + (int64_t)pair * q_stride * 32);
  const uint4* T = reinterpret_cast<const uint4*>(t + (int64_t)pair * t_stride * 32);
  const bool live = qi < nq;
  uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
  if (live) {
    qa = Q[2 * qi];
    qb = Q[2 * qi + 1];
  }
  // best two as packed keys (distance << 16 | train index): the two smallest keys are the two nearest neighbours with
  // ties resolved to the lower train index, exactly the ascending strict-< scan; update = two v_min_u32 + one v_max_u32.
  uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
  const uint8_t* mrow = mask ? mask + (int64_t)qi * nt : nullptr;
  for (int base = 0; base < nt; base += HM_TILE) {
    const int cnt = nt - base < HM_TILE ? nt - base : HM_TILE;
    __syncthreads();
    if ((int)threadIdx.x < 2 * cnt) s_t[threadIdx.x] = T[2 * base + threadIdx.x];
    __syncthreads();
    if (live) {
      for (int j = 0; j < cnt; ++j) {
        const uint32_t d = (uint32_t)popc256(qa, qb, s_t[2 * j], s_t[2 * j + 1]);
        uint32_t key = (d << 16) | (uint32_t)(base + j);
        if (mrow && !mrow[base + j]) key = 0xFFFFFFFFu;
        const uint32_t lo = min(k0, key);
        k1 = min(k1, max(k0, key));  // k0 <= k1: second smallest of {k0, k1, key}
        k0 = lo;
      }
    }
  }
  if (live) {
    const int64_t o = (int64_t)pair * out_stride + qi;
    idx0[o] = k0 == 0xFFFFFFFFu ? -1 : (int32_t)(k0 & 0xffffu);
    d0[o] = k0 == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(k0 >> 16);
    idx1[o] = k1 == 0xFFFFFFFFu ? -1 : (int32_t)(k1 & 0xffffu);
    d1[o] = k1 == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(k1 >> 16);
  }
}

// ---- matrix-core form ------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int KM_ROW = 272;  // bytes per expanded train row in LDS: 256 + 16 so that the 16-byte fragment reads of 16 rows hit 64 banks

// 4 bits -> 4 bytes of 0/1 (bit i -> byte i): (t * (1 + 2^7 + 2^14 + 2^21)) & 0x01010101
__device__ __forceinline__ uint32_t spread4(uint32_t t) { return (__umul24(t & 0xfu, 0x00204081u)) & 0x01010101u; }
__device__ __forceinline__ v4i spread16(uint32_t x) {  // 16 bits -> 16 bytes
  v4i r;
  r.x = (int)spread4(x), r.y = (int)spread4(x >> 4), r.z = (int)spread4(x >> 8), r.w = (int)spread4(x >> 12);
  return r;
}
// the same with 0xff where a bit is set (0 / -1 as signed bytes): (b << 8) - b per word, in unsigned arithmetic -- no carries between
// bytes, no signed overflow
__device__ __forceinline__ uint32_t spread4_ff(uint32_t t) {
  const uint32_t b = spread4(t);
  return (b << 8) - b;
}
__device__ __forceinline__ v4i spread16_ff(uint32_t x) {
  v4i r;
  r.x = (int)spread4_ff(x), r.y = (int)spread4_ff(x >> 4), r.z = (int)spread4_ff(x >> 8), r.w = (int)spread4_ff(x >> 12);
  return r;
}

// median of three signed integers (v_med3_i32; clang has no builtin for the integer form)
__device__ __forceinline__ int med3_i32(int a, int b, int c) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// grid: (ceil(max_query/128), pairs); 4 wavefronts, each owns 32 queries (MFMA columns); train descriptors = MFMA rows.
// C/D layout of the 32x32 shapes: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// A / B fragments of k-step s: lane (r = lane & 31, h = lane >> 5) supplies bits 32s + 16h .. +15 of train row r / query column r
// (any k order is fine as long as A and B use the same one: both come from spread16 of the same 16-bit field).
#ifndef UVO_OCC_KNN
#define UVO_OCC_KNN 6  // six workgroups per CU: 0.102 -> 0.098 ms per step; eight: 0.130 ms
#endif
__global__ __launch_bounds__(256, UVO_OCC_KNN) void k_knn2_mfma(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq_arr, int nq_fixed, int q_stride,
                                                   const uint8_t* __restrict__ t, const int32_t* __restrict__ nt_arr, int nt_fixed, int t_stride,
                                                   int out_stride, int32_t* __restrict__ idx0, uint16_t* __restrict__ d0,
                                                   int32_t* __restrict__ idx1, uint16_t* __restrict__ d1) {
  __shared__ __attribute__((aligned(16))) uint8_t s_exp[32 * KM_ROW];
  __shared__ __attribute__((aligned(16))) int32_t s_key[32];
  const int pair = blockIdx.y;
  // counts come from device arrays (an extractor's n_out may exceed the caller's capacity): never walk past a slice
  // (fixed counts were validated on the host)
  const int nq = nq_arr ? clamp_count(nq_arr[pair], q_stride < out_stride ? q_stride : out_stride) : nq_fixed;
  const int nt = nt_arr ? clamp_count(nt_arr[pair], t_stride < 65535 ? t_stride : 65535) : nt_fixed;  // the key packs the train index in 16 bits
  if (blockIdx.x * 128 >= nq) return;
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int r = lane & 31, h = lane >> 5;
  const int qi = blockIdx.x * 128 + wv * 32 + r;
  const uint32_t* Q = reinterpret_cast<const uint32_t*>(q + (int64_t)pair * q_stride * 32);
  const uint32_t* T = reinterpret_cast<const uint32_t*>(t + (int64_t)pair * t_stride * 32);
  // query fragments, held for the whole train loop (8 k-steps x 4 VGPRs) + pop(q).  The query bits are expanded to 0 / -1 (bytes 0x00 /
  // 0xff), the train bits to 0 / 1: the accumulator then holds -<q, t>, and the key of the epilogue is one shift-add away
  v4i bq[8];
  int popq = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const uint32_t w = qi < nq ? Q[(int64_t)qi * 8 + s] : 0u;
    popq += __popc(w);
    bq[s] = spread16_ff(w >> (16 * h));  // 0x01 -> 0xff per byte
  }
  int k0 = 0x7fffffff, k1 = 0x7fffffff;  // two smallest keys ((pop(t) - 2 dot) << 16 | train index), signed
  const int erow = threadIdx.x >> 3, ec = threadIdx.x & 7;  // expansion: thread -> (train row of the tile, 32-bit chunk)
  // the next tile's word is fetched while this tile is expanded and multiplied: a workgroup walks ~30 tiles one after another, and
  // without the prefetch every one of them exposes a full memory round trip
  uint32_t wnext = erow < nt ? T[(int64_t)erow * 8 + ec] : 0u;
  for (int base = 0; base < nt; base += 32) {
    __syncthreads();
    {
      const int tr = base + erow;
      const uint32_t w = wnext;
      wnext = tr + 32 < nt ? T[(int64_t)(tr + 32) * 8 + ec] : 0u;
      v4i* dst = reinterpret_cast<v4i*>(s_exp + erow * KM_ROW + ec * 32);
      dst[0] = spread16(w), dst[1] = spread16(w >> 16);
      int pc = __popc(w);  // pop(t) of the row: sum over its 8 chunk threads (adjacent lanes)
      pc += __shfl_xor(pc, 1, 64);
      pc += __shfl_xor(pc, 2, 64);
      pc += __shfl_xor(pc, 4, 64);
      if (ec == 0) s_key[erow] = tr < nt ? ((pc << 16) | tr) : 0x7fff0000;
    }
    __syncthreads();
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint8_t* arow = s_exp + r * KM_ROW + 16 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const v4i a = *reinterpret_cast<const v4i*>(arow + 32 * s);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v4i kb = *reinterpret_cast<const v4i*>(&s_key[8 * g + 4 * h]);  // rows 8g + 4h + (0..3)
      const int kbv[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // key = (pop(t) - 2 <q, t>) << 16 | train index, with acc = -<q, t>: v_lshl_add_u32; the two smallest of {k0 <= k1, key} are
        // min(k0, key) and the median of the three: v_med3_i32 + v_min_i32 -- three instructions per distance
        const int key = (int)(((uint32_t)acc[4 * g + e] << 17) + (uint32_t)kbv[e]);
        k1 = med3_i32(k0, k1, key);
        k0 = min(k0, key);
      }
    }
  }
  // the two lane halves saw different train rows of the same query: merge
  const int o0 = __shfl_xor(k0, 32, 64), o1 = __shfl_xor(k1, 32, 64);
  const int m0 = min(k0, o0), m1 = min(max(k0, o0), min(k1, o1));
  if (h == 0 && qi < nq) {
    const int64_t o = (int64_t)pair * out_stride + qi;
    const bool v0 = m0 < 0x7fff0000, v1 = m1 < 0x7fff0000;
    idx0[o] = v0 ? (m0 & 0xffff) : -1;
    d0[o] = v0 ? (uint16_t)((m0 >> 16) + popq) : (uint16_t)0xFFFF;
    idx1[o] = v1 ? (m1 & 0xffff) : -1;
    d1[o] = v1 ? (uint16_t)((m1 >> 16) + popq) : (uint16_t)0xFFFF;
  }
}

__global__ __launch_bounds__(256) void k_matrix(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                                uint16_t* __restrict__ dist) {
  __shared__ uint4 s_t[HM_TILE * 2];
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const uint4* Q = reinterpret_cast<const uint4*>(q);
  const uint4* T = reinterpret_cast<const uint4*>(t);
  const bool live = qi < nq;
  uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
  if (live) {
    qa = Q[2 * qi];
    qb = Q[2 * qi + 1];
  }
  const int base = blockIdx.y * HM_TILE;
  const int cnt = nt - base < HM_TILE ? nt - base : HM_TILE;
  if ((int)threadIdx.x < 2 * cnt) s_t[threadIdx.x] = T[2 * base + threadIdx.x];
  __syncthreads();
  if (!live) return;
  for (int j = 0; j < cnt; ++j) dist[(int64_t)qi * nt + base + j] = (uint16_t)popc256(qa, qb, s_t[2 * j], s_t[2 * j + 1]);
}

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:197-270): among the N observations of a map point take the
// descriptor whose median distance to all of them (itself included: distance 0) is smallest; median = sorted[int(0.5*(N-1))];
// first index wins ties (:257-261).  One workgroup per map point, one thread per observation (strided for N > 256).
// The k-th smallest of a row is found by bisection on the distance value (9 rounds of N popcounts) -- no sort, no storage.
__global__ __launch_bounds__(256) void k_medoid(const uint8_t* __restrict__ desc, const int32_t* __restrict__ offsets, int32_t* __restrict__ best_idx,
                                                int32_t* __restrict__ best_median) {
  __shared__ uint32_t s_best;
  const int p = blockIdx.x;
  const int o0 = offsets[p], n = offsets[p + 1] - o0;
  if (threadIdx.x == 0) s_best = 0xFFFFFFFFu;
  __syncthreads();
  if (n > 0) {
    const uint4* D = reinterpret_cast<const uint4*>(desc + (int64_t)o0 * 32);
    const int k = (int)(0.5 * (n - 1));
    for (int i = threadIdx.x; i < n; i += 256) {
      const uint4 qa = D[2 * i], qb = D[2 * i + 1];
      int lo = 0, hi = 256;  // smallest v with #{j : d(i,j) <= v} >= k+1
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int c = 0;
        for (int j = 0; j < n; ++j) c += popc256(qa, qb, D[2 * j], D[2 * j + 1]) <= mid;
        if (c >= k + 1)
          hi = mid;
        else
          lo = mid + 1;
      }
      atomicMin(&s_best, ((uint32_t)lo << 16) | (uint32_t)i);  // smallest median, then smallest index
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    best_idx[p] = n > 0 ? (int32_t)(s_best & 0xffffu) : -1;
    best_median[p] = n > 0 ? (int32_t)(s_best >> 16) : -1;
  }
}

void launch_medoid(hipStream_t s, const uint8_t* d_desc, const int32_t* d_offsets, int npoints, int32_t* d_idx, int32_t* d_med) {
  hipLaunchKernelGGL(k_medoid, dim3(npoints), dim3(256), 0, s, d_desc, d_offsets, d_idx, d_med);
}

void launch_knn2(hipStream_t s, int pairs, int max_q, const uint8_t* d_q, const int32_t* d_nq, int nq_fixed, int q_stride, const uint8_t* d_t,
                 const int32_t* d_nt, int nt_fixed, int t_stride, const uint8_t* d_mask, int out_stride, int32_t* d_idx0, uint16_t* d_d0,
                 int32_t* d_idx1, uint16_t* d_d1) {
  if (!d_mask) {
    hipLaunchKernelGGL(k_knn2_mfma, dim3((max_q + 127) / 128, pairs), dim3(256), 0, s, d_q, d_nq, nq_fixed, q_stride, d_t, d_nt, nt_fixed, t_stride,
                       out_stride, d_idx0, d_d0, d_idx1, d_d1);
    return;
  }
  hipLaunchKernelGGL(k_knn2, dim3((max_q + 255) / 256, pairs), dim3(256), 0, s, d_q, d_nq, nq_fixed, q_stride, d_t, d_nt, nt_fixed, t_stride,
                     d_mask, out_stride, d_idx0, d_d0, d_idx1, d_d1);
}

void launch_matrix(hipStream_t s, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, uint16_t* d_dist) {
  hipLaunchKernelGGL(k_matrix, dim3((nq + 255) / 256, (nt + HM_TILE - 1) / HM_TILE), dim3(256), 0, s, d_q, nq, d_t, nt, d_dist);
}

}  // namespace uvo
