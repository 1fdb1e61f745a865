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

// thread = 16 output bytes (one dwordx4 store).  Threads whose 16 columns lie inside the image copy one aligned
// dwordx4 (when the caller's rows are 16-B aligned, `vec_ok`); border / unaligned threads gather byte by byte with
// the reflected index.
__global__ __launch_bounds__(256) void k_pad_level0(const uint8_t* __restrict__ img, int w, int h, int64_t stride, int64_t frame_stride,
                                                    uint8_t* __restrict__ pyr, int64_t pyr_block, int64_t plane_off, int pitch, int ph,
                                                    int vec_ok) {
  const int qx = blockIdx.x * 64 + (threadIdx.x & 63);  // 16-byte group in the padded row
  const int py = blockIdx.y * 4 + wave_in_block();
  const int f = blockIdx.z;
  if (qx * 16 >= pitch || py >= ph) return;
  const int y = reflect101(py - kPad, h);
  const uint8_t* src = img + f * frame_stride + (int64_t)y * stride;
  const int x0 = qx * 16 - kPad;  // image column of byte 0
  uint4 v;
  if (vec_ok && x0 >= 0 && x0 + 16 <= w) {
    v = *reinterpret_cast<const uint4*>(src + x0);
  } else {
    uint32_t d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t t = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int x = reflect101(x0 + j * 4 + i, w);
        x = x < 0 ? 0 : (x >= w ? w - 1 : x);  // columns past pw (row pitch slack) stay in range
        t |= (uint32_t)src[x] << (8 * i);
      }
      d[j] = t;
    }
    v = make_uint4(d[0], d[1], d[2], d[3]);
  }
  uint8_t* dst = pyr + f * pyr_block + plane_off + (int64_t)py * pitch;
  *reinterpret_cast<uint4*>(dst + qx * 16) = v;
}

// cv::resize INTER_LINEAR, 8-bit generic path: horizontal pass in 11-bit fixed point (INTER_RESIZE_COEF_SCALE
// = 2048) into int, vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.  The coefficient tables are built
// on the host exactly as resizeGeneric_ builds them (extractor.cpp) and are indexed by *padded* output coordinates,
// i.e. the REFLECT_101 border is already folded into them: entry px of the column table holds (sx, a0, a1) of the
// level column reflect(px-16), entry py of the row table (sy0, sy1, b0, b1).
// One thread = 4 output bytes.  Interior threads fetch each of the two source rows as three aligned dwords (the four
// taps span <= 12 bytes for scale factors <= 2) and pick bytes with v_alignbyte; pad threads (reflected, decreasing
// source order) take the scalar byte path.
__device__ __forceinline__ uint32_t pick2(uint32_t d0, uint32_t d1, uint32_t d2, int o) {
  // bytes o and o+1 of the 12-byte window, in bits [0,16)
  const uint32_t lo = o < 4 ? d0 : (o < 8 ? d1 : d2);
  const uint32_t hi = o < 4 ? d1 : d2;
  return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(o & 3));
}

__global__ __launch_bounds__(256) void k_resize_level(uint8_t* __restrict__ pyr, int64_t pyr_block, int64_t src_off, int src_pitch, int sw,
                                                      int64_t dst_off, int dst_pitch, int dst_ph, int dw, int fast_ok,
                                                      const ResizeCol* __restrict__ ctab, const ResizeRow* __restrict__ rtab) {
  const int wx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int py = blockIdx.y * 4 + wave_in_block();
  const int f = blockIdx.z;
  if (wx * 4 >= dst_pitch || py >= dst_ph) return;
  const ResizeRow rr = rtab[py];
  const int b0 = rr.b0, b1 = rr.b1;
  const uint8_t* S = pyr + f * pyr_block + src_off + (int64_t)kPad * src_pitch + kPad;  // ROI origin of the source level
  const uint8_t* S0 = S + (int64_t)rr.sy0 * src_pitch;
  const uint8_t* S1 = S + (int64_t)rr.sy1 * src_pitch;
  const uint4 c01 = reinterpret_cast<const uint4*>(ctab)[wx * 2];      // columns 4wx, 4wx+1
  const uint4 c23 = reinterpret_cast<const uint4*>(ctab)[wx * 2 + 1];  // columns 4wx+2, 4wx+3
  const uint32_t cw[8] = {c01.x, c01.y, c01.z, c01.w, c23.x, c23.y, c23.z, c23.w};
  int sx[4], a0[4], a1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sx[i] = (int)(int16_t)(cw[2 * i] & 0xffff);
    a0[i] = (int)(int16_t)(cw[2 * i] >> 16);
    a1[i] = (int)(int16_t)(cw[2 * i + 1] & 0xffff);
  }
  uint32_t v = 0;
  const bool interior = fast_ok && wx * 4 >= kPad && wx * 4 + 3 < kPad + dw;  // monotone source columns
  if (interior) {
    const int base = sx[0] & ~3;
    const uint32_t* p0 = reinterpret_cast<const uint32_t*>(S0 + base);
    const uint32_t* p1 = reinterpret_cast<const uint32_t*>(S1 + base);
    const uint32_t u0 = p0[0], u1 = p0[1], u2 = p0[2];
    const uint32_t w0 = p1[0], w1 = p1[1], w2 = p1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = sx[i] - base;
      const uint32_t t0 = pick2(u0, u1, u2, o), t1 = pick2(w0, w1, w2, o);
      const int r0 = __mul24((int)(t0 & 0xff), a0[i]) + __mul24((int)((t0 >> 8) & 0xff), a1[i]);
      const int r1 = __mul24((int)(t1 & 0xff), a0[i]) + __mul24((int)((t1 >> 8) & 0xff), a1[i]);
      const int o8 = ((__mul24(b0, r0 >> 4) >> 16) + (__mul24(b1, r1 >> 4) >> 16) + 2) >> 2;  // all operands < 2^24
      v |= (uint32_t)(o8 & 0xff) << (8 * i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sx1 = sx[i] + 1 < sw ? sx[i] + 1 : sw - 1;
      const int r0 = __mul24((int)S0[sx[i]], a0[i]) + __mul24((int)S0[sx1], a1[i]);
      const int r1 = __mul24((int)S1[sx[i]], a0[i]) + __mul24((int)S1[sx1], a1[i]);
      const int o8 = ((__mul24(b0, r0 >> 4) >> 16) + (__mul24(b1, r1 >> 4) >> 16) + 2) >> 2;
      v |= (uint32_t)(o8 & 0xff) << (8 * i);
    }
  }
  uint8_t* dst = pyr + f * pyr_block + dst_off + (int64_t)py * dst_pitch;
  *reinterpret_cast<uint32_t*>(dst + wx * 4) = v;
}

void launch_pad_level0(hipStream_t s, const uint8_t* d_img, int w, int h, int64_t stride, int64_t frame_stride, uint8_t* d_pyr,
                       int64_t pyr_block, const LevelGeom& g0, int batch) {
  dim3 block(256);
  dim3 grid((g0.pitch / 16 + 63) / 64, (g0.ph + 3) / 4, batch);
  const int vec_ok = ((uintptr_t)d_img % 16 == 0 && stride % 16 == 0 && frame_stride % 16 == 0) ? 1 : 0;
  hipLaunchKernelGGL(k_pad_level0, grid, block, 0, s, d_img, w, h, stride, frame_stride, d_pyr, pyr_block, g0.plane_off, g0.pitch, g0.ph,
                     vec_ok);
}

void launch_resize_level(hipStream_t s, uint8_t* d_pyr, int64_t pyr_block, const LevelGeom& src, const LevelGeom& dst, const ResizeCol* d_ctab,
                         const ResizeRow* d_rtab, int fast_ok, int batch) {
  dim3 block(256);
  dim3 grid((dst.pitch / 4 + 63) / 64, (dst.ph + 3) / 4, batch);
  hipLaunchKernelGGL(k_resize_level, grid, block, 0, s, d_pyr, pyr_block, src.plane_off, src.pitch, src.w, dst.plane_off, dst.pitch, dst.ph,
                     dst.w, fast_ok, d_ctab, d_rtab);
}

}  // namespace uvo
