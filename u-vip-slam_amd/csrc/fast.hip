// Per-cell FAST-9/16 with in-cell 3x3 non-max suppression and the per-cell threshold fallback.
// Replaces the cell loop of ORBextractor::ComputeKeyPointsOctTree (src/ORBextractor.cc:773-812):
//   FAST(cellROI, kps, fastTh, true); if (kps.empty()) FAST(cellROI, kps, 7, true);
// (cv::FAST, TYPE_9_16: 9 contiguous circle pixels all > v+t or all < v-t; score = cornerScore<16> = the largest
// threshold that keeps the pixel a corner; NMS keeps strict 8-neighbour maxima, neighbours outside the ROI's
// 3-px-inset interior or that are not corners count as 0.)
//
// One workgroup = one cell.  The (wCell+6)x(hCell+6) ROI (<= 66x66) is staged in LDS, scores of the interior
// are computed once at t_min = min(fastTh, 7) into a second LDS plane (score is threshold independent, and a
// neighbour below the active threshold can never beat a pixel at or above it -- SURVEY.md A.3), then the cell
// decides between fastTh and 7 with one workgroup-wide vote and appends its survivors to the (frame, level)
// candidate list.  Candidate order in HBM is arbitrary: the quad-tree kernel orders by coordinates.
#include "common.hpp"

namespace uvo {

constexpr int FT_MAX = 66;      // max ROI edge: wCell < 60, + 6
constexpr int FT_PITCH = 72;

__device__ __forceinline__ int max16(const int* a) {
  int m = a[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) m = max(m, a[k]);
  return m;
}

// max over the 16 arcs of 9 contiguous ring pixels of min(d) -- sliding minimum by doubling
__device__ __forceinline__ int arc9_maxmin(const int* d) {
  int a1[16], a2[16], a4[16], a9[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a1[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) a2[k] = min(a1[k], a1[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) a4[k] = min(a2[k], a2[(k + 4) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) a9[k] = min(a4[k], d[(k + 8) & 15]);
  return max16(a9);
}

__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t* __restrict__ pyr, int64_t pyr_block, const LevelGeom* __restrict__ lv,
                                                    const CellDesc* __restrict__ cells, int fast_th, int t_min,
                                                    uint32_t* __restrict__ cand_xy, uint32_t* __restrict__ cand_sc, int64_t cand_block,
                                                    int32_t* __restrict__ cand_count, int nlevels) {
  __shared__ uint8_t s_img[FT_MAX][FT_PITCH];
  __shared__ uint8_t s_sc[FT_MAX][FT_PITCH];
  __shared__ int s_any;

  const CellDesc cd = cells[blockIdx.x];
  const int f = blockIdx.y;
  const LevelGeom& g = lv[cd.level];
  const int rw = cd.rw, rh = cd.rh;
  const uint8_t* src = pyr + f * pyr_block + g.plane_off + (int64_t)(cd.y0 + kPad) * g.pitch + (cd.x0 + kPad);
  const int tid = threadIdx.x;
  if (tid == 0) s_any = 0;
  for (int i = tid; i < rh * rw; i += 256) {
    const int r = i / rw, c = i - r * rw;
    s_img[r][c] = src[(int64_t)r * g.pitch + c];
    s_sc[r][c] = 0;
  }
  __syncthreads();

  const int iw = rw - 6, ih = rh - 6;  // interior
  if (iw <= 0 || ih <= 0) return;
  for (int i = tid; i < iw * ih; i += 256) {
    const int r = 3 + i / iw, c = 3 + i % iw;
    const int v = s_img[r][c];
    int d[16];
    d[0] = s_img[r + 3][c], d[1] = s_img[r + 3][c + 1], d[2] = s_img[r + 2][c + 2], d[3] = s_img[r + 1][c + 3];
    d[4] = s_img[r][c + 3], d[5] = s_img[r - 1][c + 3], d[6] = s_img[r - 2][c + 2], d[7] = s_img[r - 3][c + 1];
    d[8] = s_img[r - 3][c], d[9] = s_img[r - 3][c - 1], d[10] = s_img[r - 2][c - 2], d[11] = s_img[r - 1][c - 3];
    d[12] = s_img[r][c - 3], d[13] = s_img[r + 1][c - 3], d[14] = s_img[r + 2][c - 2], d[15] = s_img[r + 3][c - 1];
    uint32_t mb = 0, md = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      d[k] -= v;
      mb |= (uint32_t)(d[k] > t_min) << k;
      md |= (uint32_t)(d[k] < -t_min) << k;
    }
    // 9 contiguous set bits in the circular 16-bit mask
    auto run9 = [](uint32_t m) {
      m |= m << 16;
      uint32_t x = m & (m >> 1);
      x &= x >> 2;
      x &= x >> 4;
      x &= m >> 8;
      return (x & 0xffffu) != 0;
    };
    const bool cb = run9(mb), cdk = run9(md);
    if (cb || cdk) {
      int sb = 0, sd = 0;
      if (cb) sb = arc9_maxmin(d);
      if (cdk) {
        int nd[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) nd[k] = -d[k];
        sd = arc9_maxmin(nd);
      }
      s_sc[r][c] = (uint8_t)(max(sb, sd) - 1);
    }
  }
  __syncthreads();

  // in-cell NMS; survivors kept in registers (<= 15 interior pixels per thread: 60*60/256)
  uint32_t keep_xy[15];
  uint8_t keep_s[15];
  int nk = 0;
  bool any_hi = false;
  for (int i = tid; i < iw * ih; i += 256) {
    const int r = 3 + i / iw, c = 3 + i % iw;
    const int s = s_sc[r][c];
    if (s == 0) continue;
    const bool keep = s > s_sc[r - 1][c - 1] && s > s_sc[r - 1][c] && s > s_sc[r - 1][c + 1] && s > s_sc[r][c - 1] && s > s_sc[r][c + 1] &&
                      s > s_sc[r + 1][c - 1] && s > s_sc[r + 1][c] && s > s_sc[r + 1][c + 1];
    if (!keep) continue;
    keep_xy[nk] = (uint32_t)(c + cd.ox) | ((uint32_t)(r + cd.oy) << 16);
    keep_s[nk] = (uint8_t)s;
    ++nk;
    any_hi |= s >= fast_th;
  }
  if (any_hi) s_any = 1;
  __syncthreads();
  const int th = s_any ? fast_th : 7;
  uint32_t* out_xy = cand_xy + f * cand_block + g.cand_off;
  uint32_t* out_sc = cand_sc + f * cand_block + g.cand_off;
  int32_t* cnt = cand_count + f * nlevels + cd.level;
  for (int k = 0; k < nk; ++k) {
    if (keep_s[k] >= th) {
      const int pos = atomicAdd(cnt, 1);
      if (pos < g.cand_cap) {
        out_xy[pos] = keep_xy[k];
        out_sc[pos] = keep_s[k];
      }
    }
  }
}

void launch_fast_cells(hipStream_t s, const uint8_t* d_pyr, int64_t pyr_block, const LevelGeom* d_lv, const CellDesc* d_cells, int total_cells,
                       int fast_th, uint32_t* d_cand_xy, uint32_t* d_cand_sc, int64_t cand_block, int32_t* d_cand_count, int nlevels,
                       int batch) {
  const int t_min = fast_th < 7 ? fast_th : 7;
  hipLaunchKernelGGL(k_fast_cells, dim3(total_cells, batch), dim3(256), 0, s, d_pyr, pyr_block, d_lv, d_cells, fast_th, t_min, d_cand_xy,
                     d_cand_sc, cand_block, d_cand_count, nlevels);
}

}  // namespace uvo
