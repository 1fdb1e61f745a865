// 7x7 sigma=2 Gaussian blur of every pyramid level into the "blurred" pyramid.
// Replaces cv::GaussianBlur(work, work, Size(7,7), 2, 2, BORDER_REFLECT_101) at src/ORBextractor.cc:942.
//
// Semantics kept: the reference blurs the ROI of the padded buffer in place and non-isolated,
// so border taps read the real, un-blurred REFLECT_101 pad, and after the call the pad still holds un-blurred
// pixels which computeOrbDescriptor samples up to 2 px deep.  Here the blur is out of place: the output plane
// holds the blurred interior plus a 4-px ring copied from the un-blurred pad -- all the descriptor can reach.
// Arithmetic is OpenCV's symmetric-smooth integer engine: taps round(g*256) per pass (18,34,49,55,49,34,18),
// row pass u8 -> int, column pass (sum + 2^15) >> 16 saturated to u8.
//
// One workgroup = 64x16 output pixels.  The 72x22 source window is staged once in LDS with aligned dword
// loads, the row pass result (<= 65535, kept as u16) goes back to LDS, and each thread then emits one dword.
#include "common.hpp"

namespace uvo {

constexpr int GT_W = 64, GT_H = 16;

__global__ __launch_bounds__(256) void k_gauss7(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int64_t pyr_block,
                                                const LevelGeom* __restrict__ lv, int nlevels, int4 taps) {
  __shared__ __attribute__((aligned(16))) uint8_t s_src[GT_H + 6][GT_W + 8];
  __shared__ __attribute__((aligned(16))) uint16_t s_row[GT_H + 6][GT_W];

  // tile -> level
  int level = 0, tile = blockIdx.x;
  int tx_n = 0;
  for (;; ++level) {
    tx_n = (lv[level].w + 8 + GT_W - 1) / GT_W;
    const int ty_n = (lv[level].h + 8 + GT_H - 1) / GT_H;
    if (tile < tx_n * ty_n || level == nlevels - 1) break;
    tile -= tx_n * ty_n;
  }
  const LevelGeom g = lv[level];
  const int f = blockIdx.y;
  const int ox = -4 + (tile % tx_n) * GT_W;  // tile origin in ROI coordinates
  const int oy = -4 + (tile / tx_n) * GT_H;
  const uint8_t* src = pyr + f * pyr_block + g.plane_off;
  uint8_t* dst = blur + f * pyr_block + g.plane_off;
  const int tid = threadIdx.x;

  // stage rows oy-3 .. oy+GT_H+2, columns ox-4 .. ox+GT_W+3 (dword aligned: ROI origin is at byte 16 of a 64-B pitched row)
  for (int i = tid; i < (GT_H + 6) * ((GT_W + 8) / 4); i += 256) {
    const int r = i / ((GT_W + 8) / 4), c4 = i % ((GT_W + 8) / 4);
    int py = oy - 3 + r + kPad;
    py = py < 0 ? 0 : (py >= g.ph ? g.ph - 1 : py);
    int px = ox - 4 + c4 * 4 + kPad;
    px = px < 0 ? 0 : (px > g.pitch - 4 ? g.pitch - 4 : px);
    *reinterpret_cast<uint32_t*>(&s_src[r][c4 * 4]) = *reinterpret_cast<const uint32_t*>(src + (int64_t)py * g.pitch + px);
  }
  __syncthreads();
  // row pass: s_row[r][c] = sum_i k[i] * src(ox + c - 3 + i) ; source column ox+c-3+i sits at s_src[r][c+1+i]
  for (int i = tid; i < (GT_H + 6) * GT_W; i += 256) {
    const int r = i / GT_W, c = i % GT_W;
    const uint8_t* p = &s_src[r][c + 1];
    const int s = taps.x * (p[0] + p[6]) + taps.y * (p[1] + p[5]) + taps.z * (p[2] + p[4]) + taps.w * p[3];
    s_row[r][c] = (uint16_t)s;
  }
  __syncthreads();
  // column pass, 4 pixels per thread
  const int r = tid / 16, c0 = (tid % 16) * 4;
  const int y = oy + r;
  if (y >= g.h + 4) return;
  uint32_t out = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + i;
    const int x = ox + c;
    int v;
    if (x >= 0 && x < g.w && y >= 0 && y < g.h) {
      const int s = taps.x * (s_row[r][c] + s_row[r + 6][c]) + taps.y * (s_row[r + 1][c] + s_row[r + 5][c]) +
                    taps.z * (s_row[r + 2][c] + s_row[r + 4][c]) + taps.w * s_row[r + 3][c];
      v = (s + (1 << 15)) >> 16;
      v = v > 255 ? 255 : v;
    } else {
      v = s_src[r + 3][c + 4];  // pad ring: un-blurred copy
    }
    out |= (uint32_t)v << (8 * i);
  }
  const int x0 = ox + c0;
  if (x0 < g.w + 4) *reinterpret_cast<uint32_t*>(dst + (int64_t)(y + kPad) * g.pitch + (x0 + kPad)) = out;
}

void launch_gauss7(hipStream_t s, const uint8_t* d_pyr, uint8_t* d_blur, int64_t pyr_block, const LevelGeom* d_lv, const Geom& g, int4 taps,
                   int batch) {
  int tiles = 0;
  for (int l = 0; l < g.nlevels; ++l) tiles += ((g.lv[l].w + 8 + GT_W - 1) / GT_W) * ((g.lv[l].h + 8 + GT_H - 1) / GT_H);
  hipLaunchKernelGGL(k_gauss7, dim3(tiles, batch), dim3(256), 0, s, d_pyr, d_blur, pyr_block, d_lv, g.nlevels, taps);
}

}  // namespace uvo
