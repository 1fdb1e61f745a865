// 7x7 sigma=2 Gaussian blur of every pyramid level into the "blurred" pyramid.
// Replaces cv::GaussianBlur(work, work, Size(7,7), 2, 2, BORDER_REFLECT_101) at src/ORBextractor.cc:942.
//
// Semantics kept: the reference blurs the ROI of the padded buffer in place and non-isolated,
// so border taps read the real, un-blurred REFLECT_101 pad, and after the call the pad still holds un-blurred
// pixels which computeOrbDescriptor samples up to 2 px deep.  Here the blur is out of place: the output plane
// holds the blurred interior plus a 4-px ring copied from the un-blurred pad -- all the descriptor can reach.
// The output plane is TILED: 16 x 8-pixel tiles of 128 bytes (one cache line), tile (ty, tx) at ((ty * pitch / 16) + tx) * 128 of
// the plane.  Its only reader is k_describe, whose 37 x 40-byte windows touch 18.7 lines this way instead of 48 (describe.hip).
// Arithmetic is OpenCV's symmetric-smooth integer engine: taps round(g*256) per pass (18,34,49,55,49,34,18),
// row pass u8 -> int, column pass (sum + 2^15) >> 16 saturated to u8.  The column pass runs on the fp32 pipe and is still exact:
// row sums (< 2^16) times taps (sum 257) stay below 2^24 wherever the result is not saturated, fp32 add / fma issue in 2.4
// cycles per wavefront on gfx950 where the 24-bit integer multiply takes 4.4 (tools/ubench/valu_rate3.hip), and with the
// wavefront's fp32 rounding mode set to round-toward-zero v_cvt_pk_u8_f32 is floor + saturate + byte insert in one instruction.
//
// Streaming stencil without LDS: one wavefront owns a vertical strip of 64 lanes x 4 pixels (lanes 0 and 63 are halo,
// 248 useful columns) and walks down the rows.  Per row a lane issues ONE aligned dword load, takes its neighbours'
// dwords by cross-lane shifts, does the row pass for its 4 pixels in registers and keeps the last 7 row-pass results in
// a register ring (the loop is unrolled by 7 so ring slots are compile-time); the column pass then emits one dword.
// Every byte of a level is loaded once per strip segment (+6 halo rows per GS_ROWS) and written once.
#include "common.hpp"
#include "fast_geom.hpp"
#include "gauss_body.hpp"

namespace uvo {

template <bool SSE2>
__global__ __launch_bounds__(256) void k_gauss7(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int64_t pyr_block,
                                                const LevelGeom* __restrict__ lv, int nlevels, int4 taps, int rows_per_seg, Level0View l0, GaussPlans plans) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tile[4][GS_TILE_DW];
  gauss7_body<SSE2>((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y, s_tile, pyr, blur, pyr_block, lv, nlevels, taps, rows_per_seg, l0, plans);
}

#ifndef UVO_GAUSS_ROWS
#define UVO_GAUSS_ROWS 64
#endif
int gauss7_rows_per_seg(int batch) { return batch >= 16 ? UVO_GAUSS_ROWS : 16; }  // fewer, longer segments when the batch already fills the chip (6 halo rows are re-read per segment)
GaussPlans gauss7_plans(const Geom& g, int rows_per_seg) {
  GaussPlans P;
  for (int l = 0; l < kMaxLevels; ++l) {
    P.p[l] = StripPlan{};
    if (l < g.nlevels) fast_strip_plan(g.lv[l].w + 8, g.lv[l].h + 8, rows_per_seg, P.p[l]);
  }
  return P;
}
int gauss7_blocks_per_frame(const Geom& g, int rows_per_seg) {
  int items = 0;
  for (int l = 0; l < g.nlevels; ++l) {
    StripPlan plan;
    fast_strip_plan(g.lv[l].w + 8, g.lv[l].h + 8, rows_per_seg, plan);
    items += plan.items;
  }
  return (items + 3) / 4;
}

void launch_gauss7(hipStream_t s, const uint8_t* d_pyr, uint8_t* d_blur, int64_t pyr_block, const LevelGeom* d_lv, const Geom& g, int4 taps,
                   int batch, int sse2_rounding, Level0View l0) {
  const int rows_per_seg = gauss7_rows_per_seg(batch), bx = gauss7_blocks_per_frame(g, rows_per_seg);
  if (sse2_rounding)
    hipLaunchKernelGGL(k_gauss7<true>, dim3(bx, batch), dim3(256), 0, s, d_pyr, d_blur, pyr_block, d_lv, g.nlevels, taps, rows_per_seg, l0, gauss7_plans(g, rows_per_seg));
  else
    hipLaunchKernelGGL(k_gauss7<false>, dim3(bx, batch), dim3(256), 0, s, d_pyr, d_blur, pyr_block, d_lv, g.nlevels, taps, rows_per_seg, l0, gauss7_plans(g, rows_per_seg));
}

}  // namespace uvo
