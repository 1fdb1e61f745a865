// Per-cell FAST-9/16 with in-cell 3x3 non-max suppression and the per-cell threshold fallback.
// Replaces the cell loop of ORBextractor::ComputeKeyPointsOctTree (src/ORBextractor.cc:773-812):
//   FAST(cellROI, kps, fastTh, true); if (kps.empty()) FAST(cellROI, kps, 7, true);
// (cv::FAST, TYPE_9_16: 9 contiguous circle pixels all > v+t or all < v-t; score = cornerScore<16> = the largest
// threshold that keeps the pixel a corner; NMS keeps strict 8-neighbour maxima, neighbours outside the ROI's
// 3-px-inset interior or that are not corners count as 0.)
//
// One workgroup = one cell, three phases over an LDS copy of the (wCell+6)x(hCell+6) ROI (<= 66x66):
//   1. screen : every interior pixel, 4 per lane from aligned LDS dwords, with the two cheapest necessary conditions
//               (any 9-arc contains ring pixel 0 or 8, and 4 or 12, with one polarity); survivors (a few %) are
//               compacted into an LDS list -- the expensive test never runs on diverged, mostly idle waves;
//   2. score  : full segment test + cornerScore for the listed pixels, once, at t_min = min(fastTh, 7) (the score is
//               threshold independent and a neighbour below the active threshold can never beat a pixel at or above
//               it -- SURVEY.md A.3), into a zero-initialised LDS score plane;
//   3. select : in-cell NMS on the score plane, one workgroup vote between fastTh and the literal 7, append to the
//               (frame, level) candidate list.  Candidate order in HBM is arbitrary: the quad-tree orders by coordinates.
#include "common.hpp"

namespace uvo {

constexpr int FT_MAX = 66;      // max ROI edge: wCell < 60, + 6
constexpr int FT_DW = 19;       // LDS row pitch in dwords (76 B >= 3 + 66 + 3)
constexpr int FT_PITCH = FT_DW * 4;
constexpr int FT_LIST = 60 * 60;

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
  __shared__ uint32_t s_img32[FT_MAX * FT_DW];
  __shared__ uint32_t s_sc32[FT_MAX * FT_DW];
  __shared__ uint16_t s_list[FT_LIST];
  __shared__ int s_nlist, s_any;
  const uint8_t* s_img = reinterpret_cast<const uint8_t*>(s_img32);
  uint8_t* s_sc = reinterpret_cast<uint8_t*>(s_sc32);

  const CellDesc cd = cells[blockIdx.x];
  const int f = blockIdx.y;
  const LevelGeom& g = lv[cd.level];
  const int rw = cd.rw, rh = cd.rh;
  const int tid = threadIdx.x;
  // ROI origin in the padded plane; rows are 64-B pitched and plane offsets 256-B aligned, so (x & ~3) is dword aligned
  const int px0 = cd.x0 + kPad;
  const int a = px0 & 3;
  const uint8_t* src = pyr + f * pyr_block + g.plane_off + (int64_t)(cd.y0 + kPad) * g.pitch + (px0 - a);
  const int ndw = (a + rw + 3) >> 2;
  if (tid == 0) {
    s_nlist = 0;
    s_any = 0;
  }
  for (int i = tid; i < rh * ndw; i += 256) {
    const int r = i / ndw, d = i - r * ndw;
    s_img32[r * FT_DW + d] = *reinterpret_cast<const uint32_t*>(src + (int64_t)r * g.pitch + d * 4);
  }
  for (int i = tid; i < rh * FT_DW; i += 256) s_sc32[i] = 0;
  __syncthreads();

  const int iw = rw - 6, ih = rh - 6;  // interior
  if (iw <= 0 || ih <= 0) return;

  // ---- phase 1: screen, 4 pixels (one LDS dword) per work item ----
  const int g0 = (a + 3) >> 2, g1 = (a + rw - 4) >> 2;  // dword columns that touch the interior
  const int ngr = g1 - g0 + 1;
  for (int i = tid; i < ih * ngr; i += 256) {
    const int r = 3 + i / ngr, gq = g0 + i % ngr;
    const uint32_t* row = &s_img32[r * FT_DW];
    const uint32_t C = row[gq];
    const uint32_t L = gq > 0 ? row[gq - 1] : 0u, R = gq + 1 < FT_DW ? row[gq + 1] : 0u;
    const uint32_t U = s_img32[(r - 3) * FT_DW + gq], D = s_img32[(r + 3) * FT_DW + gq];
    const uint32_t P12 = (L >> 8) | (C << 24);  // bytes x-3
    const uint32_t P4 = (C >> 24) | (R << 8);   // bytes x+3
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = gq * 4 + k - a;  // ROI column
      const int v = (C >> (8 * k)) & 0xff;
      const int p0 = (D >> (8 * k)) & 0xff, p8 = (U >> (8 * k)) & 0xff, p4 = (P4 >> (8 * k)) & 0xff, p12 = (P12 >> (8 * k)) & 0xff;
      const int hi = v + t_min, lo = v - t_min;
      const bool bright = ((p0 > hi) | (p8 > hi)) & ((p4 > hi) | (p12 > hi));
      const bool dark = ((p0 < lo) | (p8 < lo)) & ((p4 < lo) | (p12 < lo));
      const bool pass = (bright | dark) & (c >= 3) & (c < rw - 3);
      // wave-aggregated append
      const uint64_t m = __ballot(pass);
      if (m) {
        const int lane = tid & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_nlist, __popcll(m));
        base = __shfl(base, 0, 64);
        if (pass) s_list[base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)((r << 8) | c);
      }
    }
  }
  __syncthreads();
  const int nlist = s_nlist;

  // ---- phase 2: full segment test + score for the screened pixels ----
  for (int i = tid; i < nlist; i += 256) {
    const int rc = s_list[i];
    const int r = rc >> 8, c = rc & 0xff;
    const uint8_t* p = s_img + r * FT_PITCH + a + c;
    const int v = p[0];
    int d[16];
    d[0] = p[3 * FT_PITCH], d[1] = p[3 * FT_PITCH + 1], d[2] = p[2 * FT_PITCH + 2], d[3] = p[FT_PITCH + 3];
    d[4] = p[3], d[5] = p[-FT_PITCH + 3], d[6] = p[-2 * FT_PITCH + 2], d[7] = p[-3 * FT_PITCH + 1];
    d[8] = p[-3 * FT_PITCH], d[9] = p[-3 * FT_PITCH - 1], d[10] = p[-2 * FT_PITCH - 2], d[11] = p[-FT_PITCH - 3];
    d[12] = p[-3], d[13] = p[FT_PITCH - 3], d[14] = p[2 * FT_PITCH - 2], d[15] = p[3 * FT_PITCH - 1];
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
      s_sc[r * FT_PITCH + c] = (uint8_t)(max(sb, sd) - 1);
    }
  }
  __syncthreads();

  // ---- phase 3: in-cell NMS; survivors kept in registers (<= ceil(3600/256) = 15 per thread) ----
  uint32_t keep_xy[15];
  uint8_t keep_s[15];
  int nk = 0;
  bool any_hi = false;
  for (int i = tid; i < nlist; i += 256) {
    const int rc = s_list[i];
    const int r = rc >> 8, c = rc & 0xff;
    const uint8_t* q = s_sc + r * FT_PITCH + c;
    const int s = q[0];
    if (s == 0) continue;
    const bool keep = s > q[-FT_PITCH - 1] && s > q[-FT_PITCH] && s > q[-FT_PITCH + 1] && s > q[-1] && s > q[1] && s > q[FT_PITCH - 1] &&
                      s > q[FT_PITCH] && s > q[FT_PITCH + 1];
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
