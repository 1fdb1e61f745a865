// Pyramid build: level-0 border pad and the per-level bilinear down-scale with reflected pad.
// Replaces ORBextractor::ComputePyramid (src/ORBextractor.cc:963-1004): cv::copyMakeBorder(REFLECT_101) at
// level 0 (:996) and cv::resize(INTER_LINEAR) + copyMakeBorder(REFLECT_101|ISOLATED) at levels >= 1 (:982,:988).
//
// HBM-bound streaming kernels: every thread produces 4 consecutive bytes of one padded output row (one
// aligned dword store, rows are 64-B pitched).  Pad pixels are produced by evaluating the level at the
// reflected coordinate, so each level is written exactly once and no second border pass exists.
#include "common.hpp"
#include "pyr_schedule.hpp"

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
  // 16 columns that lie wholly in the left or right pad are a reversed run of 16 image columns (REFLECT_101): four
  // unaligned dword loads and a byte swap each, instead of sixteen reflected byte gathers that stall the whole wavefront
  const int r0 = x0 + 16 <= 0 ? -(x0 + 15) : 2 * (w - 1) - x0 - 15;  // first image column of that run
  if (vec_ok && x0 >= 0 && x0 + 16 <= w) {
    v = *reinterpret_cast<const uint4*>(src + x0);
  } else if ((x0 + 16 <= 0 || x0 >= w) && r0 >= 0 && r0 + 16 <= w) {
    uint32_t d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t t;
      __builtin_memcpy(&t, src + r0 + 4 * j, 4);
      d[3 - j] = __builtin_bswap32(t);
    }
    v = make_uint4(d[0], d[1], d[2], d[3]);
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
// level column reflect(px-16) with the weights scaled by 16, entry py of the row table (sy0, sy1, b0, b1).
// One thread = 4 output bytes x RZ_ROWS rows.  Each of the two source rows is fetched as three aligned dwords (the four tap
// pairs of a dword lie inside 12 bytes for scale factors up to ~1.33; checked per level when the tables are built); one
// v_perm_b32 with a per-thread selector gathers the four left taps, a second one on the window shifted by a byte the four
// right taps.  Levels that do not fit the window gather bytes instead.
// The column weights are stored scaled by 16 (a0 << 4, a1 << 4), so that a row sum comes out as r << 4 and `(r >> 4) << 8`
// -- the operand v_mul_hi_u32_u24 needs to give (b * (r >> 4)) >> 16 in one instruction -- is a single AND.
__device__ __forceinline__ uint32_t mulhi24(uint32_t a, uint32_t b) {  // v_mul_hi_u32_u24: bits [47:32] of the 48-bit product
  return (uint32_t)(((uint64_t)(a & 0xffffffu) * (uint64_t)(b & 0xffffffu)) >> 32);
}
// left / right tap bytes (bits [8i, 8i+8) of L, R) -> ((left*a0 + right*a1) >> 4) << 8
template <int I>
__device__ __forceinline__ uint32_t hrow(uint32_t L, uint32_t R, uint32_t a0s, uint32_t a1s) {
  const uint32_t r16 = __umul24((L >> (8 * I)) & 0xffu, a0s) + __umul24((R >> (8 * I)) & 0xffu, a1s);
  return r16 & 0xffffff00u;
}
__device__ __forceinline__ uint32_t vrow(uint32_t q0, uint32_t q1, uint32_t b0s, uint32_t b1s) {
  return ((mulhi24(b0s, q0) + mulhi24(b1s, q1) + 2u) >> 2) & 0xffu;  // ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2
}

constexpr int RZ_ROWS = 4;  // output rows per thread: one column-table fetch, RZ_ROWS x 2 independent row fetches in flight

#ifndef UVO_OCC_RESIZE
#define UVO_OCC_RESIZE 6  // six workgroups per CU (80 VGPRs): 0.208 -> 0.197 ms per step; eight (64 VGPRs) spill: 0.263 ms
#endif
__global__ __launch_bounds__(256, UVO_OCC_RESIZE) void k_resize_level(uint8_t* __restrict__ pyr, int64_t pyr_block, int64_t src_off, int src_pitch, int sw,
                                                      int64_t dst_off, int dst_pitch, int dst_ph, int dw, int fast_ok,
                                                      const ResizeCol* __restrict__ ctab, const ResizeRow* __restrict__ rtab,
                                                      uint32_t nwx_magic, Level0View l0, int sh, uint32_t per_frame, uint32_t per_frame_magic, uint32_t per_xcd,
                                                      int batch, uint32_t nwx, int wx0, int rg0, int row_end) {
  // flat index -> (row group, dword column): rows are a few dozen to 150 dwords long, so a (64 x rows) tiling would leave up to
  // a third of the lanes idle on some levels.  gid / nwx by multiply-high with ceil(2^32 / nwx) (exact for gid < 2^20).
  // (an XCD walks whole frames: neighbouring workgroups read the same source rows -- dealt round-robin, every XCD's L2 fetched its own
  // copy of them: 1.39 x the algorithmic bytes per launch)
  // workgroup b runs on XCD b & 7: it takes item (b & 7) * per_xcd + (b >> 3) of the frame-major list of (frame, workgroup of the frame)
  const uint32_t vb = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (vb >= per_frame * (uint32_t)batch) return;
  const int f = (int)__umulhi(vb, per_frame_magic);  // vb / per_frame (exact: vb * per_frame < 2^32)
  const uint32_t gid = (vb - (uint32_t)f * per_frame) * 256u + threadIdx.x;
  // the launch covers dword columns wx0 .. wx0 + nwx - 1 and rows rg0 * 4 .. row_end - 1 of the padded plane: all of it, or the ROI and the
  // four pixels around it that anything ever reads (launch_resize_level)
  const uint32_t rgl = __umulhi(gid, nwx_magic);
  const int wx = (int)(gid - rgl * nwx) + wx0;
  const uint32_t rg = rgl + (uint32_t)rg0;
  const int py0 = (int)rg * RZ_ROWS;
  if (py0 >= row_end) return;
  dst_ph = row_end;
  // ROI origin of the source level; level 1 may read the caller's image in place (l0.vbase: the resize only looks at the ROI)
  const bool ip = l0.vbase != nullptr;
  if (ip) src_pitch = l0.pitch;
  const uint8_t* S = ip ? l0.vbase + f * l0.frame_stride + (int64_t)kPad * src_pitch + kPad : pyr + f * pyr_block + src_off + (int64_t)kPad * src_pitch + kPad;
  // The 12-byte tap window of the last columns reaches up to 8 bytes past the ROI's last pixel (taps of weight zero).  Inside a padded
  // plane those bytes are the pad; in place they are the next row, and behind the last row of the last frame nothing the caller owns: the
  // window of that row's last lanes is pulled back into the frame (the bytes it then misses carry no weight: see `edge` below).
  const int64_t s_end = ip ? (int64_t)(sh - 1) * src_pitch + sw : INT64_MAX;
  const uint4 c01 = reinterpret_cast<const uint4*>(ctab)[wx * 2];      // columns 4wx, 4wx+1
  const uint4 c23 = reinterpret_cast<const uint4*>(ctab)[wx * 2 + 1];  // columns 4wx+2, 4wx+3
  const uint32_t cw[8] = {c01.x, c01.y, c01.z, c01.w, c23.x, c23.y, c23.z, c23.w};
  uint32_t sx[4], a0[4], a1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sx[i] = cw[2 * i] & 0xffffu;
    a0[i] = cw[2 * i] >> 16;
    a1[i] = cw[2 * i + 1] & 0xffffu;
  }
  // window base and tap selector come ready-made from the table (pad fields of the four column entries); they cover the reflected
  // pad columns too (taps in decreasing order), so every thread of a level takes the same path
  const bool interior = fast_ok != 0;
  const uint32_t base = cw[1] >> 16;
  const uint32_t sel = (cw[3] >> 16) | (cw[5] & 0xffff0000u);
  // the row group's four table entries in two 16-byte loads (the table is padded to whole groups), then all eight source rows at once:
  // nothing here waits for anything but the two table fetches
  static_assert(RZ_ROWS == 4 && sizeof(ResizeRow) == 8, "the row group is read as 2 x uint4");
  const uint4 t01 = reinterpret_cast<const uint4*>(rtab)[rg * 2], t23 = reinterpret_cast<const uint4*>(rtab)[rg * 2 + 1];
  const uint32_t tw[8] = {t01.x, t01.y, t01.z, t01.w, t23.x, t23.y, t23.z, t23.w};
  ResizeRow rr[RZ_ROWS];
  uint32_t u[RZ_ROWS][3], w[RZ_ROWS][3];
#pragma unroll
  for (int j = 0; j < RZ_ROWS; ++j) {
    rr[j].sy0 = (int16_t)(tw[2 * j] & 0xffffu), rr[j].sy1 = (int16_t)(tw[2 * j] >> 16);
    rr[j].b0 = (int16_t)(tw[2 * j + 1] & 0xffffu), rr[j].b1 = (int16_t)(tw[2 * j + 1] >> 16);
  }
  // (in place, only a row group that reaches the frame's last source row has to look at the frame's end)
  const bool guard = ip && max(max((int)rr[0].sy1, (int)rr[1].sy1), max((int)rr[2].sy1, (int)rr[3].sy1)) >= sh - 1;
  if (interior && !guard) {
#pragma unroll
    for (int j = 0; j < RZ_ROWS; ++j) {
      const uint32_t* p0 = reinterpret_cast<const uint32_t*>(S + (int64_t)rr[j].sy0 * src_pitch + base);
      const uint32_t* p1 = reinterpret_cast<const uint32_t*>(S + (int64_t)rr[j].sy1 * src_pitch + base);
      u[j][0] = p0[0], u[j][1] = p0[1], u[j][2] = p0[2];
      w[j][0] = p1[0], w[j][1] = p1[1], w[j][2] = p1[2];
    }
  } else if (interior) {
#pragma unroll
    for (int j = 0; j < RZ_ROWS; ++j) {
      const int64_t o0 = (int64_t)rr[j].sy0 * src_pitch + base, o1 = (int64_t)rr[j].sy1 * src_pitch + base;
      const uint32_t* p0 = reinterpret_cast<const uint32_t*>(S + o0);
      const uint32_t* p1 = reinterpret_cast<const uint32_t*>(S + o1);
      // (a window dword behind the frame's last byte is not loaded: it holds no tap of non-zero weight; rows and the frame end are
      // dword-aligned in this mode)
      u[j][0] = p0[0], u[j][1] = o0 + 4 < s_end ? p0[1] : 0u, u[j][2] = o0 + 8 < s_end ? p0[2] : 0u;
      w[j][0] = p1[0], w[j][1] = o1 + 4 < s_end ? p1[1] : 0u, w[j][2] = o1 + 8 < s_end ? p1[2] : 0u;
    }
  }
  uint32_t vout[RZ_ROWS];
#pragma unroll
  for (int j = 0; j < RZ_ROWS; ++j) {
    uint32_t L0, R0, L1, R1;  // left / right taps of the four columns in the two source rows
    if (interior) {
      L0 = __builtin_amdgcn_perm(u[j][1], u[j][0], sel);
      R0 = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(u[j][2], u[j][1], 1), __builtin_amdgcn_alignbyte(u[j][1], u[j][0], 1), sel);
      L1 = __builtin_amdgcn_perm(w[j][1], w[j][0], sel);
      R1 = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(w[j][2], w[j][1], 1), __builtin_amdgcn_alignbyte(w[j][1], w[j][0], 1), sel);
    } else {
      const uint8_t* S0 = S + (int64_t)rr[j].sy0 * src_pitch;
      const uint8_t* S1 = S + (int64_t)rr[j].sy1 * src_pitch;
      L0 = R0 = L1 = R1 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t sx1 = sx[i] + 1 < (uint32_t)sw ? sx[i] + 1 : (uint32_t)sw - 1;
        L0 |= (uint32_t)S0[sx[i]] << (8 * i), R0 |= (uint32_t)S0[sx1] << (8 * i);
        L1 |= (uint32_t)S1[sx[i]] << (8 * i), R1 |= (uint32_t)S1[sx1] << (8 * i);
      }
    }
    const uint32_t b0s = (uint32_t)rr[j].b0 << 8, b1s = (uint32_t)rr[j].b1 << 8;
    uint32_t v = vrow(hrow<0>(L0, R0, a0[0], a1[0]), hrow<0>(L1, R1, a0[0], a1[0]), b0s, b1s);
    v |= vrow(hrow<1>(L0, R0, a0[1], a1[1]), hrow<1>(L1, R1, a0[1], a1[1]), b0s, b1s) << 8;
    v |= vrow(hrow<2>(L0, R0, a0[2], a1[2]), hrow<2>(L1, R1, a0[2], a1[2]), b0s, b1s) << 16;
    v |= vrow(hrow<3>(L0, R0, a0[3], a1[3]), hrow<3>(L1, R1, a0[3], a1[3]), b0s, b1s) << 24;
    vout[j] = v;
  }
  // all four rows of a group exist except in the plane's last group: one test per thread instead of one per row
  uint8_t* dst = pyr + f * pyr_block + dst_off + (int64_t)py0 * dst_pitch + wx * 4;
  if (py0 + RZ_ROWS <= dst_ph) {
#pragma unroll
    for (int j = 0; j < RZ_ROWS; ++j) *reinterpret_cast<uint32_t*>(dst + (uint32_t)(j * dst_pitch)) = vout[j];
  } else {
#pragma unroll
    for (int j = 0; j < RZ_ROWS; ++j)
      if (py0 + j < dst_ph) *reinterpret_cast<uint32_t*>(dst + (uint32_t)(j * dst_pitch)) = vout[j];
  }
}

void launch_pad_level0(hipStream_t s, const uint8_t* d_img, int w, int h, int64_t stride, int64_t frame_stride, uint8_t* d_pyr,
                       int64_t pyr_block, const LevelGeom& g0, int batch) {
  dim3 block(256);
  dim3 grid((g0.pitch / 16 + 63) / 64, (g0.ph + 3) / 4, batch);
  const int vec_ok = ((uintptr_t)d_img % 16 == 0 && stride % 16 == 0 && frame_stride % 16 == 0) ? 1 : 0;
  hipLaunchKernelGGL(k_pad_level0, grid, block, 0, s, d_img, w, h, stride, frame_stride, d_pyr, pyr_block, g0.plane_off, g0.pitch, g0.ph,
                     vec_ok);
}

// ring = 0: the whole padded plane, as cv::copyMakeBorder of src/ORBextractor.cc:988 leaves it.  ring = 4 (the extractor's hot path): the ROI and
// the four pixels around it -- no stage reads further out (FAST and the orientation patch stay inside the ROI, the blur reaches 3 pixels and
// copies 4 into the blurred plane's ring, the next level's resize reads the ROI): a sixth fewer pixels to interpolate and to write.
void launch_resize_level(hipStream_t s, uint8_t* d_pyr, int64_t pyr_block, const LevelGeom& src, const LevelGeom& dst, const ResizeCol* d_ctab,
                         const ResizeRow* d_rtab, int fast_ok, int batch, Level0View l0, int ring) {
  dim3 block(256);
  uint32_t nwx = (uint32_t)dst.pitch / 4, groups = ((uint32_t)dst.ph + RZ_ROWS - 1) / RZ_ROWS;
  int wx0 = 0, rg0 = 0, row_end = dst.ph;
  if (ring > 0 && ring < kPad && (kPad - ring) % 4 == 0) {
    wx0 = (kPad - ring) / 4, rg0 = (kPad - ring) / RZ_ROWS, row_end = dst.h + kPad + ring;
    nwx = (uint32_t)((dst.w + kPad + ring + 3) / 4 - wx0), groups = (uint32_t)((row_end + RZ_ROWS - 1) / RZ_ROWS - rg0);
  }
  const uint32_t magic = (uint32_t)((0x100000000ull + nwx - 1) / nwx);
  const uint32_t per_frame = (nwx * groups + 255) / 256, per_xcd = (per_frame * (uint32_t)batch + 7) / 8;
  const uint32_t pf_magic = (uint32_t)((0x100000000ull + per_frame - 1) / per_frame);
  hipLaunchKernelGGL(k_resize_level, dim3(8 * per_xcd), block, 0, s, d_pyr, pyr_block, src.plane_off, src.pitch, src.w, dst.plane_off, dst.pitch, dst.ph,
                     dst.w, fast_ok, d_ctab, d_rtab, magic, l0, src.h, per_frame, pf_magic, per_xcd, batch, nwx, wx0, rg0, row_end);
}


// =====================================================================================================================
// k_pyramid: the whole pyramid of a batch in ONE launch (schedule: pyr_schedule.hpp).
// Workgroup = (band, frame).  Every wavefront keeps a few ROLES -- (level, 64-lane column chunk) -- for the whole band: the column
// tables of its lanes and the horizontal pass of the last source row it has seen live in registers.  Per macro-step the workgroup
// reads what every level does from a table (output rows [lo, hi), new source rows [k_lo, k_hi]), every wavefront streams its roles
// over those rows, and the workgroup meets at a barrier: rows written before the barrier are read behind it by the same workgroup
// (same CU: one L1, one L2) -- no LDS, no agent-scope fences, every level is written once and read back out of the L2 it was just
// written through.  Rows are addressed as buffer resource (the plane) + per-lane byte offset + scalar row offset: no vector address
// arithmetic per row.
//
// Arithmetic of a resize role (cv::resize INTER_LINEAR, 8-bit generic path), on the fp32 pipe with the wavefront's
// rounding mode at round-toward-zero, every step exact:
//   horizontal  r16 = L * (a0 << 4) + R * (a1 << 4)            two exact products, sum < 2^24
//               q   = r >> 4 = floor(r16 / 256)                fma(r16, 2^-8, 2^23) - 2^23: the fma's single rounding truncates at unit size
//   vertical    t0' = fma(q0, b0 / 65536, 2^23)                = 2^23 + ((b0 * q0) >> 16)
//               w   = fma(q1, b1 / 65536, t0')                 = 2^23 + t0 + t1   (t0' is an integer: the truncation only hits q1 * b1)
//               o'  = fma(w, 1/4, 2^23 - 2^21 + 1/2)           = 2^23 + ((t0 + t1 + 2) >> 2): the byte is the low mantissa byte
// A source row's horizontal pass is computed once and used by the (one or two) output rows that read it.
struct PyrLevel {
  int w, h, pitch, ctab_off;
  int64_t plane_off;
  int rtab_off, pad;
};
struct PyrLevels {
  PyrLevel l[kMaxLevels];
};

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pyr_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);  // raw buffer, 32-bit data format
}

struct PyrConst {
  float k23, k256, kq, kfin;
};

// per-slot state of a resize role
struct PyrResize {
  float a0[4], a1[4];      // (a << 4) of the lane's four output columns, as floats
  uint32_t sx[4];          // byte-gather path only: left tap columns
  uint32_t sel, base;      // tap selector / first byte of the lane's 12-byte source window
  uint32_t voff_dst;       // byte offset of the lane's dword in a padded destination row
  float Hp[4];             // q of the last source row seen (the upper tap of the next output row)
  bool active;
};

template <bool FAST>
__device__ __forceinline__ void pyr_resize_init(PyrResize& R, const ResizeCol* __restrict__ ctab, int nwx, int chunk, int lane) {
  const int wx = chunk * 64 + lane;
  const int wxc = wx < nwx ? wx : nwx - 1;
  const uint4 c01 = reinterpret_cast<const uint4*>(ctab)[wxc * 2];      // columns 4wx, 4wx+1
  const uint4 c23 = reinterpret_cast<const uint4*>(ctab)[wxc * 2 + 1];  // columns 4wx+2, 4wx+3
  const uint32_t cw[8] = {c01.x, c01.y, c01.z, c01.w, c23.x, c23.y, c23.z, c23.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    R.sx[i] = cw[2 * i] & 0xffffu;
    R.a0[i] = (float)(cw[2 * i] >> 16);
    R.a1[i] = (float)(cw[2 * i + 1] & 0xffffu);
    asm volatile("" : "+v"(R.a0[i]), "+v"(R.a1[i]));  // keep them floats: the products below stay on the fp32 pipe
    R.Hp[i] = 0.f;
  }
  R.base = cw[1] >> 16;
  R.sel = (cw[3] >> 16) | (cw[5] & 0xffff0000u);
  R.voff_dst = (uint32_t)wx * 4u;
  R.active = wx < nwx;
}

// what a resize role fetches in one step: the new source rows (at most kPyrMaxSrcRows; past the last new row the index is clamped, so
// the surplus slots repeat the last row: unconditional loads, and slot 6 always ends up holding the last row seen) and the vertical
// weights of the step's eight slots (every lane reads the same words: they arrive in vector registers, no scalar-to-vector moves)
struct PyrFetch {
  uint32_t u[kPyrMaxSrcRows][3];
};
template <bool FAST>
__device__ __forceinline__ void pyr_resize_fetch(PyrFetch& F, const PyrResize& R, __amdgpu_buffer_rsrc_t src, int src_pitch, int sw, int k_lo, int nsrc) {
#pragma unroll
  for (int j = 0; j < kPyrMaxSrcRows; ++j) {
    const int kr = k_lo + (j < nsrc ? j : nsrc - 1);
    const int soff = kr * src_pitch;
    if (FAST) {
      const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(src, (int)R.base, soff, 0);
      F.u[j][0] = v.x, F.u[j][1] = v.y, F.u[j][2] = v.z;
    } else {
      uint32_t Lb = 0, Rb = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t sx1 = R.sx[i] + 1 < (uint32_t)sw ? R.sx[i] + 1 : (uint32_t)sw - 1;
        Lb |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(src, (int)R.sx[i], soff, 0) << (8 * i);
        Rb |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(src, (int)sx1, soff, 0) << (8 * i);
      }
      F.u[j][0] = Lb, F.u[j][1] = Rb, F.u[j][2] = 0;
    }
  }
}

template <bool FAST>
__device__ __forceinline__ void pyr_hpass(float* H, const uint32_t* u, const PyrResize& R, const PyrConst& C) {
  uint32_t L4, R4;
  if (FAST) {
    L4 = __builtin_amdgcn_perm(u[1], u[0], R.sel);
    R4 = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(u[2], u[1], 1), __builtin_amdgcn_alignbyte(u[1], u[0], 1), R.sel);
  } else {
    L4 = u[0], R4 = u[1];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float Lf = (float)((L4 >> (8 * i)) & 0xffu), Rf = (float)((R4 >> (8 * i)) & 0xffu);  // v_cvt_f32_ubyte<i>
    const float r16 = __builtin_fmaf(Rf, R.a1[i], Lf * R.a0[i]);
    H[i] = __builtin_fmaf(r16, C.k256, C.k23) - C.k23;
  }
}

// one output row from the q rows of its two taps (Hu: upper, Hl: lower); soff / dual = kPyrNoStore: the hardware drops the store
__device__ __forceinline__ void pyr_emit(const PyrResize& R, __amdgpu_buffer_rsrc_t dst, const float* Hu, const float* Hl, float b0, float b1, int soff, int dual,
                                         const PyrConst& C) {
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t0 = __builtin_fmaf(Hu[i], b0, C.k23);
    const float w = __builtin_fmaf(Hl[i], b1, t0);
    o[i] = __builtin_fmaf(w, C.kq, C.kfin);
  }
  const uint32_t v = __builtin_amdgcn_perm(f2u(o[1]), f2u(o[0]), 0x0c0c0400u) | __builtin_amdgcn_perm(f2u(o[3]), f2u(o[2]), 0x04000c0cu);
  // lanes past the row's last dword carry an offset outside the resource too
  __builtin_amdgcn_raw_buffer_store_b32(v, dst, (int)R.voff_dst, soff, 0);
  __builtin_amdgcn_raw_buffer_store_b32(v, dst, (int)R.voff_dst, dual, 0);  // the pad row that repeats this one
}

// Streams the step's source-row slots, straight-line: slot j's q row, the output row it completes (upper tap = the slot before; slot 0:
// the row carried from the step before), two stores.
template <bool FAST>
__device__ __forceinline__ void pyr_resize_run(const PyrFetch& F, PyrResize& R, __amdgpu_buffer_rsrc_t dst, const PyrStepLevel* __restrict__ T, uint32_t vzero,
                                               const int* soff, const int* dual, uint32_t flags, const PyrConst& C) {
  // the vertical weights of the step's eight slots: every lane reads the same words, so they arrive in vector registers (an fma that
  // reads a scalar register issues at half the rate, and a scalar-to-vector move per weight costs as much)
  float b[16];
  const uint4* bw = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(T) + 16 + vzero);  // (vzero: an opaque per-lane 0 -- vector loads)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 w = bw[q];
    b[4 * q] = u2f(w.x), b[4 * q + 1] = u2f(w.y), b[4 * q + 2] = u2f(w.z), b[4 * q + 3] = u2f(w.w);
  }
  float H[kPyrMaxSrcRows + 1][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) H[0][i] = R.Hp[i];
#pragma unroll
  for (int j = 0; j < kPyrMaxSrcRows; ++j) {
    pyr_hpass<FAST>(H[j + 1], F.u[j], R, C);
    if (j == 0 && (flags & 1u)) {  // the level's clamped first row: both taps on its own source row
      asm volatile("");
#pragma unroll
      for (int i = 0; i < 4; ++i) H[0][i] = H[1][i];
    }
    pyr_emit(R, dst, H[j], H[j + 1], b[2 * j], b[2 * j + 1], soff[j], dual[j], C);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) R.Hp[i] = H[kPyrMaxSrcRows][i];  // the last row seen (surplus slots repeat it)
  if (flags & 2u) pyr_emit(R, dst, R.Hp, R.Hp, b[14], b[15], soff[7], dual[7], C);  // the level's clamped last row
}

// level 0: image rows [lo, hi) -> padded plane (cv::copyMakeBorder REFLECT_101, src/ORBextractor.cc:996); lane = 16 bytes of a padded row.
// Every lane whose 16 columns lie wholly inside the image or wholly inside a pad takes the same path: four dword loads at per-lane
// addresses (ascending inside the image, descending for a reflected run) and one byte permute each (identity or reversal) -- no
// divergent branch between a row's loads, so all rows of a step are in flight together.  Lanes that straddle an image edge (only
// when the width is no multiple of 16) gather bytes.
struct PyrCopy {
  int kind;            // 0: four dwords, 2: byte gather, 3: no column
  int off[4];          // byte offset of output dword m inside an image row
  uint32_t sel;        // v_perm selector: identity or byte reversal
  int x0;
  uint32_t voff_dst;
};
__device__ __forceinline__ void pyr_copy_init(PyrCopy& K, int w, int pitch, int chunk, int lane) {
  const int qx = chunk * 64 + lane;
  K.x0 = qx * 16 - kPad;  // image column of byte 0
  const int r0 = K.x0 + 16 <= 0 ? -(K.x0 + 15) : 2 * (w - 1) - K.x0 - 15;  // first image column of the reversed run of a whole-pad group
  const bool inside = K.x0 >= 0 && K.x0 + 16 <= w, padrun = (K.x0 + 16 <= 0 || K.x0 >= w) && r0 >= 0 && r0 + 16 <= w;
  K.kind = qx * 16 >= pitch ? 3 : ((inside || padrun) ? 0 : 2);
  K.sel = padrun ? 0x00010203u : 0x03020100u;
#pragma unroll
  for (int m = 0; m < 4; ++m) K.off[m] = padrun ? r0 + 12 - 4 * m : (inside ? K.x0 + 4 * m : 0);
  K.voff_dst = (uint32_t)qx * 16u;
}
constexpr int kPyrCopyLoads = kPyrMaxSrcRows * 4;  // vector loads a copy wavefront issues per step (constant: the store drain counts past them)
__device__ __forceinline__ void pyr_copy_fetch(uint32_t (*v)[4], const PyrCopy& K, const uint8_t* __restrict__ imgf, int64_t stride, int lo, int hi) {
  const int nrows = hi - lo;
#pragma unroll
  for (int j = 0; j < kPyrMaxSrcRows; ++j) {
    const int y = lo + (j < nrows ? j : (nrows > 0 ? nrows - 1 : 0));
    const uint8_t* src = imgf + (int64_t)y * stride;
#pragma unroll
    for (int m = 0; m < 4; ++m) __builtin_memcpy(&v[j][m], src + K.off[m], 4);  // (dword loads need no alignment on this target)
  }
}
__device__ __forceinline__ void pyr_copy_store(const uint32_t (*v)[4], const PyrCopy& K, const uint8_t* __restrict__ imgf, int w, int64_t stride,
                                               __amdgpu_buffer_rsrc_t dst, int pitch, int h, int lo, int hi) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int nrows = hi - lo;
#pragma unroll
  for (int j = 0; j < kPyrMaxSrcRows; ++j) {
    if (j < nrows) {
      const int y = lo + j;
      u32x4 d;
      if (K.kind == 2) {  // a group that straddles an image edge: bytes through the reflected index
        const uint8_t* src = imgf + (int64_t)y * stride;
        uint32_t t[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          int x = reflect101(K.x0 + k, w);
          x = x < 0 ? 0 : (x >= w ? w - 1 : x);  // columns past the padded width (row pitch slack) stay in range
          t[k >> 2] |= (uint32_t)src[x] << (8 * (k & 3));
        }
        d = u32x4{t[0], t[1], t[2], t[3]};
      } else {
        d = u32x4{__builtin_amdgcn_perm(v[j][0], v[j][0], K.sel), __builtin_amdgcn_perm(v[j][1], v[j][1], K.sel), __builtin_amdgcn_perm(v[j][2], v[j][2], K.sel),
                  __builtin_amdgcn_perm(v[j][3], v[j][3], K.sel)};
      }
      if (K.kind != 3) {
        __builtin_amdgcn_raw_buffer_store_b128(d, dst, (int)K.voff_dst, (kPad + y) * pitch, 0);
        if (y >= 1 && y <= kPad) __builtin_amdgcn_raw_buffer_store_b128(d, dst, (int)K.voff_dst, (kPad - y) * pitch, 0);  // top pad: row -y = row y
        if (y >= h - 1 - kPad && y <= h - 2) __builtin_amdgcn_raw_buffer_store_b128(d, dst, (int)K.voff_dst, (kPad + 2 * (h - 1) - y) * pitch, 0);
      }
    }
  }
}

template <int NW, int NSLOT, bool FAST>
__global__ __launch_bounds__(64 * NW, 4) void k_pyramid(const uint8_t* __restrict__ img, int64_t stride, int64_t frame_stride, uint8_t* __restrict__ pyr,
                                                     int64_t pyr_block, PyrLevels LV, int nlevels, const PyrRole* __restrict__ roles,
                                                     const PyrStepLevel* __restrict__ steps, const int32_t* __restrict__ band_step,
                                                     const ResizeCol* __restrict__ ctab) {
  const int band = blockIdx.x, f = blockIdx.y;
  const int wv = wave_in_block(), lane = threadIdx.x & 63;
  const int s0 = band_step[band], s1 = band_step[band + 1];
  uint8_t* pf = pyr + f * pyr_block;
  const uint32_t rw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(roles)[wv * NSLOT]);
  if ((rw0 & 0xffu) == 0u) {
    // ---- a copy wavefront: image rows -> padded level-0 plane.  The rows of step s + 1 are fetched while those of step s are stored:
    // image rows depend on nothing, so the HBM round trip never sits between two barriers.  The barrier needs this wavefront's STORES to
    // have landed, not its prefetch: it waits until all but the youngest kPyrCopyLoads vector-memory operations are done. ----
    const uint8_t* imgf = img + f * frame_stride;
    const PyrLevel g = LV.l[0];
    PyrCopy K;
    pyr_copy_init(K, g.w, g.pitch, (int)(rw0 >> 16), lane);
    const __amdgpu_buffer_rsrc_t dst = pyr_rsrc(pf + g.plane_off, (uint32_t)(g.pitch * (g.h + 2 * kPad)));
    auto rows_of = [&](int s, int& lo, int& hi) {
      const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(steps + (int64_t)s * nlevels)[0]);
      lo = (int)(int16_t)(a & 0xffffu), hi = lo + (int)(int16_t)(a >> 16);
    };
    uint32_t cur[kPyrMaxSrcRows][4], nxt[kPyrMaxSrcRows][4];
    int lo, hi, lo_n = 0, hi_n = 0;
    rows_of(s0, lo, hi);
    pyr_copy_fetch(cur, K, imgf, stride, lo, hi);
    for (int s = s0; s < s1; ++s) {
      if (s + 1 < s1) rows_of(s + 1, lo_n, hi_n);
      else lo_n = hi_n = lo;
      pyr_copy_store(cur, K, imgf, g.w, stride, dst, g.pitch, g.h, lo, hi);
      asm volatile("" ::: "memory");  // the prefetch stays behind the stores: the wait below counts on that order
      pyr_copy_fetch(nxt, K, imgf, stride, lo_n, hi_n);
      asm volatile("" ::: "memory");
      static_assert(kPyrCopyLoads == 28, "the immediate of the wait below");
      asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < kPyrMaxSrcRows; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) cur[j][m] = nxt[j][m];
      lo = lo_n, hi = hi_n;
    }
    return;
  }
  __builtin_amdgcn_s_setreg(0x801, 3);  // MODE.fp_round (fp32) = toward zero: see the arithmetic above
  PyrConst C = {8388608.0f, 1.0f / 256.0f, 0.25f, 6291456.5f};
  asm volatile("" : "+v"(C.k23), "+v"(C.k256), "+v"(C.kq), "+v"(C.kfin));
  // ---- a resize wavefront: its roles ----
  int level[NSLOT];
  PyrResize R[NSLOT];
  __amdgpu_buffer_rsrc_t rs_src[NSLOT], rs_dst[NSLOT];
  int src_pitch[NSLOT], sw[NSLOT];
  uint32_t vzero = 0;
  asm volatile("" : "+v"(vzero));
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const uint32_t rw = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(roles)[wv * NSLOT + i]);
    level[i] = (int)(rw & 0xffu);
    const int chunk = (int)(rw >> 16);
    const int l = level[i] == kPyrNop ? 1 : level[i];
    const PyrLevel g = LV.l[l], gs = LV.l[l - 1];
    src_pitch[i] = gs.pitch, sw[i] = gs.w;
    rs_dst[i] = pyr_rsrc(pf + g.plane_off, (uint32_t)(g.pitch * (g.h + 2 * kPad)));
    // the source resource starts at the ROI origin of the level below and ends with its plane
    rs_src[i] = pyr_rsrc(pf + gs.plane_off + (int64_t)kPad * gs.pitch + kPad, (uint32_t)(gs.pitch * (gs.h + kPad) - kPad));
    pyr_resize_init<FAST>(R[i], ctab + g.ctab_off, g.pitch >> 2, chunk, lane);
    if (!R[i].active) R[i].voff_dst = 0x3fffff00u;  // outside every resource: the hardware drops the store
  }
  for (int s = s0; s < s1; ++s) {
    const PyrStepLevel* T[NSLOT];
    int k_lo[NSLOT], nsrc[NSLOT];
    uint32_t flags[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      T[i] = steps + (int64_t)s * nlevels + (level[i] == kPyrNop ? 0 : level[i]);
      const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(T[i])[0]);
      flags[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(T[i])[2]);
      k_lo[i] = (int)(int16_t)(a & 0xffffu), nsrc[i] = level[i] == kPyrNop ? 0 : (int)(int16_t)(a >> 16);
    }
    if (NSLOT <= 2 && NW <= 8) {
      // every slot's rows in flight before the first one is touched
      PyrFetch F[NSLOT];
#pragma unroll
      for (int i = 0; i < NSLOT; ++i)
        if (nsrc[i] > 0) pyr_resize_fetch<FAST>(F[i], R[i], rs_src[i], src_pitch[i], sw[i], k_lo[i], nsrc[i]);
#pragma unroll
      for (int i = 0; i < NSLOT; ++i)
        if (nsrc[i] > 0) {
          int so[8], du[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) so[j] = __builtin_amdgcn_readfirstlane(T[i]->soff[j]), du[j] = __builtin_amdgcn_readfirstlane(T[i]->dual[j]);
          pyr_resize_run<FAST>(F[i], R[i], rs_dst[i], T[i], vzero, so, du, flags[i], C);
        }
    } else {
#pragma unroll
      for (int i = 0; i < NSLOT; ++i)
        if (nsrc[i] > 0) {
          PyrFetch F;
          int so[8], du[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) so[j] = __builtin_amdgcn_readfirstlane(T[i]->soff[j]), du[j] = __builtin_amdgcn_readfirstlane(T[i]->dual[j]);
          pyr_resize_fetch<FAST>(F, R[i], rs_src[i], src_pitch[i], sw[i], k_lo[i], nsrc[i]);
          pyr_resize_run<FAST>(F, R[i], rs_dst[i], T[i], vzero, so, du, flags[i], C);
        }
    }
    __syncthreads();
  }
}


// =====================================================================================================================
// k_pyr_stream: one level of the pyramid as a streaming launch -- the form the large levels take (thousands of independent
// wavefronts: latency is hidden by occupancy and by register prefetch, no barrier anywhere).  A wavefront owns a 64-lane column chunk
// of the level and a run of consecutive BLOCKS of kPyrMaxSrcRows source rows (pyr_build_blocks): per block the same straight-line code
// as a step of k_pyramid, from the same kind of control words; the rows of the next block are in flight while the current one is
// computed.  Level 1 reads the caller's image in place (a buffer resource of exactly the frame's bytes: the few bytes a tap window
// reaches past the last row read as zero and carry weight zero); in that launch the copy of the image into the padded level-0 plane
// rides along as extra work items.
template <bool FAST>
__global__ __launch_bounds__(256, 4) void k_pyr_stream(PyrStreamArgs A) {
  const int lane = threadIdx.x & 63;
  const int blocks_per_frame = (A.items_per_frame + 3) >> 2;
  const int vb = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);  // an XCD walks whole frames: neighbouring runs share their rows in one L2
  const int f = vb / blocks_per_frame;
  int item = (vb - f * blocks_per_frame) * 4 + wave_in_block();
  if (item >= A.items_per_frame) return;
  uint8_t* pf = A.pyr + f * A.pyr_block;
  const int nres = A.nchunks * A.nsegs;
  if (item >= nres) {
    // ---- a copy item: image rows [r0, r1) of one 64-lane column chunk into the padded level-0 plane ----
    item -= nres;
    const int chunk = item % A.copy_chunks, seg = item / A.copy_chunks;
    const int r0 = seg * A.copy_rows_per_item, r1 = min(r0 + A.copy_rows_per_item, A.img_h);
    const uint8_t* imgf = A.img + f * A.img_frame_stride;
    PyrCopy K;
    pyr_copy_init(K, A.img_w, A.l0_pitch, chunk, lane);
    const __amdgpu_buffer_rsrc_t dst = pyr_rsrc(pf + A.l0_plane_off, (uint32_t)(A.l0_pitch * (A.img_h + 2 * kPad)));
    uint32_t ca[kPyrMaxSrcRows][4], cb[kPyrMaxSrcRows][4];
    pyr_copy_fetch(ca, K, imgf, A.img_stride, r0, min(r0 + kPyrMaxSrcRows, r1));
    for (int r = r0; r < r1; r += 2 * kPyrMaxSrcRows) {
      const int ra = min(r + kPyrMaxSrcRows, r1), rb = min(ra + kPyrMaxSrcRows, r1), rc = min(rb + kPyrMaxSrcRows, r1);
      pyr_copy_fetch(cb, K, imgf, A.img_stride, min(ra, r1 - 1), max(rb, min(ra, r1 - 1) + 1));
      pyr_copy_store(ca, K, imgf, A.img_w, A.img_stride, dst, A.l0_pitch, A.img_h, r, ra);
      if (ra < r1) {
        pyr_copy_fetch(ca, K, imgf, A.img_stride, min(rb, r1 - 1), max(rc, min(rb, r1 - 1) + 1));
        pyr_copy_store(cb, K, imgf, A.img_w, A.img_stride, dst, A.l0_pitch, A.img_h, ra, rb);
      }
    }
    return;
  }
  // ---- a resize item: blocks [b0, b1) of one column chunk ----
  const int chunk = item % A.nchunks, seg = item / A.nchunks;
  const int b0 = seg * A.blocks_per_item, b1 = min(b0 + A.blocks_per_item, A.nblocks);
  if (b0 >= b1) return;
  __builtin_amdgcn_s_setreg(0x801, 3);  // MODE.fp_round (fp32) = toward zero: the arithmetic of k_pyramid
  PyrConst C = {8388608.0f, 1.0f / 256.0f, 0.25f, 6291456.5f};
  asm volatile("" : "+v"(C.k23), "+v"(C.k256), "+v"(C.kq), "+v"(C.kfin));
  uint32_t vzero = 0;
  asm volatile("" : "+v"(vzero));
  const __amdgpu_buffer_rsrc_t rs_src = pyr_rsrc(A.src + f * A.src_frame_stride + A.src_origin, A.src_bytes);
  const __amdgpu_buffer_rsrc_t rs_dst = pyr_rsrc(pf + A.dst_plane_off, (uint32_t)(A.dst_pitch * (A.dst_h + 2 * kPad)));
  PyrResize R;
  pyr_resize_init<FAST>(R, A.ctab, A.dst_pitch >> 2, chunk, lane);
  if (!R.active) R.voff_dst = 0x3fffff00u;  // outside every resource: the hardware drops the store
  PyrFetch Fa, Fb;
  auto fetch = [&](PyrFetch& F, int b) {
    const int bc = min(b, b1 - 1);  // (past the run: a repeat of its last block, never used)
    const int k_lo = bc * kPyrMaxSrcRows;
    pyr_resize_fetch<FAST>(F, R, rs_src, A.src_pitch, A.sw, k_lo, min(kPyrMaxSrcRows, A.sh - k_lo));
  };
  auto run = [&](const PyrFetch& F, int b) {
    const PyrStepLevel* T = A.blocks + b;
    int so[8], du[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) so[j] = __builtin_amdgcn_readfirstlane(T->soff[j]), du[j] = __builtin_amdgcn_readfirstlane(T->dual[j]);
    const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)T->flags);
    pyr_resize_run<FAST>(F, R, rs_dst, T, vzero, so, du, flags, C);
  };
  fetch(Fa, b0);
  if (b0 > 0) {  // the row in front of the run: the upper tap of the first block's slot 0
    uint32_t u[3];
    const int soff = (b0 * kPyrMaxSrcRows - 1) * A.src_pitch;
    if (FAST) {
      const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs_src, (int)R.base, soff, 0);
      u[0] = v.x, u[1] = v.y, u[2] = v.z;
    } else {
      uint32_t Lb = 0, Rb = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t sx1 = R.sx[i] + 1 < (uint32_t)A.sw ? R.sx[i] + 1 : (uint32_t)A.sw - 1;
        Lb |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs_src, (int)R.sx[i], soff, 0) << (8 * i);
        Rb |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs_src, (int)sx1, soff, 0) << (8 * i);
      }
      u[0] = Lb, u[1] = Rb, u[2] = 0;
    }
    pyr_hpass<FAST>(R.Hp, u, R, C);
  }
  for (int b = b0; b < b1; b += 2) {
    fetch(Fb, b + 1);
    run(Fa, b);
    if (b + 1 < b1) {
      fetch(Fa, b + 2);
      run(Fb, b + 1);
    }
  }
}

template <int NW, int NSLOT>
static void launch_pyramid_t(hipStream_t s, dim3 grid, bool fast, const uint8_t* d_img, int64_t stride, int64_t frame_stride, uint8_t* d_pyr,
                             int64_t pyr_block, const PyrLevels& LV, int nlevels, const PyrPlanDev& plan, const ResizeCol* d_ctab) {
  if (fast)
    hipLaunchKernelGGL((k_pyramid<NW, NSLOT, true>), grid, dim3(64 * NW), 0, s, d_img, stride, frame_stride, d_pyr, pyr_block, LV, nlevels, plan.d_roles,
                       plan.d_steps, plan.d_band_step, d_ctab);
  else
    hipLaunchKernelGGL((k_pyramid<NW, NSLOT, false>), grid, dim3(64 * NW), 0, s, d_img, stride, frame_stride, d_pyr, pyr_block, LV, nlevels, plan.d_roles,
                       plan.d_steps, plan.d_band_step, d_ctab);
}

// workgroup shapes the kernel is built for: (wavefronts, role slots per resize wavefront); every copy role takes a wavefront of its own
#define UVO_PYR_SHAPES(X) X(4, 2) X(8, 2) X(8, 3) X(16, 2) X(16, 3) X(16, 4) X(16, 6) X(16, 8)
bool pyr_shape_for_roles(int nresize, int ncopy, int min_waves, int& nwaves, int& nslots) {
#define UVO_PYR_TRY(NW, NS)                                                     \
  if (NW >= min_waves && NW > ncopy && nresize <= (NW - ncopy) * NS) {          \
    nwaves = NW, nslots = NS;                                                   \
    return true;                                                                \
  }
  UVO_PYR_SHAPES(UVO_PYR_TRY)
#undef UVO_PYR_TRY
  return false;
}

int launch_pyramid(hipStream_t s, const uint8_t* d_img, int64_t stride, int64_t frame_stride, uint8_t* d_pyr, int64_t pyr_block, const Geom& g,
                   const int* fast_ok, const PyrPlanDev& plan, const ResizeCol* d_ctab, int batch) {
  PyrLevels LV;
  bool fast = true;
  for (int l = 0; l < kMaxLevels; ++l) LV.l[l] = PyrLevel{0, 0, 0, 0, 0, 0, 0};
  for (int l = 0; l < g.nlevels; ++l) {
    LV.l[l] = PyrLevel{g.lv[l].w, g.lv[l].h, g.lv[l].pitch, g.lv[l].xtab_off, g.lv[l].plane_off, 0, 0};
    if (l > 0 && !fast_ok[l]) fast = false;  // one level whose taps do not fit the 12-byte window: every level gathers bytes
  }
  dim3 grid(plan.nbands, batch);
#define UVO_PYR_CASE(NW, NS)                                                                                                                        \
  if (plan.nwaves == NW && plan.nslots == NS) {                                                                                                     \
    launch_pyramid_t<NW, NS>(s, grid, fast, d_img, stride, frame_stride, d_pyr, pyr_block, LV, g.nlevels, plan, d_ctab);            \
    return 0;                                                                                                                                       \
  }
  UVO_PYR_SHAPES(UVO_PYR_CASE)
#undef UVO_PYR_CASE
  return -1;
}


void launch_pyr_stream(hipStream_t s, const PyrStreamArgs& A, bool fast, int batch) {
  const int blocks_per_frame = (A.items_per_frame + 3) / 4;
  if (fast)
    hipLaunchKernelGGL(k_pyr_stream<true>, dim3(blocks_per_frame * batch), dim3(256), 0, s, A);
  else
    hipLaunchKernelGGL(k_pyr_stream<false>, dim3(blocks_per_frame * batch), dim3(256), 0, s, A);
}


// development probe: one wavefront that does nothing for `us` microseconds (what does pure latency in a lane cost the step?)
__global__ void k_probe_delay(int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
}
void launch_probe_delay(hipStream_t s, int us) { hipLaunchKernelGGL(k_probe_delay, dim3(1), dim3(64), 0, s, us); }

}  // namespace uvo
