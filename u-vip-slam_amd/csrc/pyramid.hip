// Pyramid build: level-0 border pad and the per-level bilinear down-scale with reflected pad.
// Replaces ORBextractor::ComputePyramid (src/ORBextractor.cc:963-1004): cv::copyMakeBorder(REFLECT_101) at
// level 0 (:996) and cv::resize(INTER_LINEAR) + copyMakeBorder(REFLECT_101|ISOLATED) at levels >= 1 (:982,:988).
//
// HBM-bound streaming kernels: every thread produces 4 consecutive bytes of one padded output row (one
// aligned dword store, rows are 64-B pitched).  Pad pixels are produced by evaluating the level at the
// reflected coordinate, so each level is written exactly once and no second border pass exists.
#include "common.hpp"

namespace uvo {

__device__ __forceinline__ int reflect101(int p, int len) {
  // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba ; pad (16) is always smaller than len here
  p = p < 0 ? -p : p;
  p = p >= len ? 2 * (len - 1) - p : p;
  return p;
}

__global__ __launch_bounds__(256) void k_pad_level0(const uint8_t* __restrict__ img, int w, int h, int64_t stride, int64_t frame_stride,
                                                    uint8_t* __restrict__ pyr, int64_t pyr_block, int64_t plane_off, int pitch, int ph) {
  const int wx = blockIdx.x * blockDim.x + threadIdx.x;  // dword index in the padded row
  const int py = blockIdx.y;
  const int f = blockIdx.z;
  if (wx * 4 >= pitch) return;
  const int y = reflect101(py - kPad, h);
  const uint8_t* src = img + f * frame_stride + (int64_t)y * stride;
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int x = reflect101(wx * 4 + i - kPad, w);
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);  // columns past pw (row pitch slack) stay in range
    v |= (uint32_t)src[x] << (8 * i);
  }
  uint8_t* dst = pyr + f * pyr_block + plane_off + (int64_t)py * pitch;
  *reinterpret_cast<uint32_t*>(dst + wx * 4) = v;
  (void)ph;
}

// cv::resize INTER_LINEAR, 8-bit generic path: horizontal pass in 11-bit fixed point (INTER_RESIZE_COEF_SCALE
// = 2048) into int, vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.  The coefficient tables
// (xofs/ialpha, yofs/ibeta) are built on the host exactly as resizeGeneric_ builds them (extractor.cpp).
__global__ __launch_bounds__(256) void k_resize_level(uint8_t* __restrict__ pyr, int64_t pyr_block, int64_t src_off, int src_pitch, int sw,
                                                      int sh, int64_t dst_off, int dst_pitch, int dw, int dh,
                                                      const int32_t* __restrict__ xofs, const int16_t* __restrict__ xalpha,
                                                      const int32_t* __restrict__ yofs, const int16_t* __restrict__ ybeta) {
  const int wx = blockIdx.x * blockDim.x + threadIdx.x;
  const int py = blockIdx.y;
  const int f = blockIdx.z;
  if (wx * 4 >= dst_pitch) return;
  const int y = reflect101(py - kPad, dh);
  int sy0 = yofs[y], sy1 = sy0 + 1;
  sy0 = sy0 < 0 ? 0 : (sy0 >= sh ? sh - 1 : sy0);
  sy1 = sy1 < 0 ? 0 : (sy1 >= sh ? sh - 1 : sy1);
  const int b0 = ybeta[2 * y], b1 = ybeta[2 * y + 1];
  const uint8_t* S = pyr + f * pyr_block + src_off + (int64_t)kPad * src_pitch + kPad;  // ROI origin of the source level
  const uint8_t* S0 = S + (int64_t)sy0 * src_pitch;
  const uint8_t* S1 = S + (int64_t)sy1 * src_pitch;
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int x = reflect101(wx * 4 + i - kPad, dw);
    x = x < 0 ? 0 : (x >= dw ? dw - 1 : x);
    const int sx = xofs[x];
    const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
    const int a0 = xalpha[2 * x], a1 = xalpha[2 * x + 1];
    const int r0 = S0[sx] * a0 + S0[sx1] * a1;
    const int r1 = S1[sx] * a0 + S1[sx1] * a1;
    const int o = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v |= (uint32_t)(o & 0xff) << (8 * i);
  }
  uint8_t* dst = pyr + f * pyr_block + dst_off + (int64_t)py * dst_pitch;
  *reinterpret_cast<uint32_t*>(dst + wx * 4) = v;
}

void launch_pad_level0(hipStream_t s, const uint8_t* d_img, int w, int h, int64_t stride, int64_t frame_stride, uint8_t* d_pyr,
                       int64_t pyr_block, const LevelGeom& g0, int batch) {
  dim3 block(256);
  dim3 grid((g0.pitch / 4 + 255) / 256, g0.ph, batch);
  hipLaunchKernelGGL(k_pad_level0, grid, block, 0, s, d_img, w, h, stride, frame_stride, d_pyr, pyr_block, g0.plane_off, g0.pitch, g0.ph);
}

void launch_resize_level(hipStream_t s, uint8_t* d_pyr, int64_t pyr_block, const LevelGeom& src, const LevelGeom& dst, const int32_t* d_xofs,
                         const int16_t* d_xalpha, const int32_t* d_yofs, const int16_t* d_ybeta, int batch) {
  dim3 block(256);
  dim3 grid((dst.pitch / 4 + 255) / 256, dst.ph, batch);
  hipLaunchKernelGGL(k_resize_level, grid, block, 0, s, d_pyr, pyr_block, src.plane_off, src.pitch, src.w, src.h, dst.plane_off, dst.pitch,
                     dst.w, dst.h, d_xofs, d_xalpha, d_yofs, d_ybeta);
}

}  // namespace uvo
