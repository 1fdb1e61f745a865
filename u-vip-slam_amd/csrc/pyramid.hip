// Pyramid build: level-0 border pad and the per-level bilinear down-scale with reflected pad.
// Replaces ORBextractor::ComputePyramid (src/ORBextractor.cc:963-1004): cv::copyMakeBorder(REFLECT_101) at
// level 0 (:996) and cv::resize(INTER_LINEAR) + copyMakeBorder(REFLECT_101|ISOLATED) at levels >= 1 (:982,:988).
//
// HBM-bound streaming kernels: every thread produces 4 consecutive bytes of one padded output row (one
// aligned dword store, rows are 64-B pitched).  Pad pixels are produced by evaluating the level at the
// reflected coordinate, so each level is written exactly once and no second border pass exists.
#include "common.hpp"
#include <algorithm>
#include <cstring>
#include <mutex>

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
  const uint32_t per_frame = (nwx * groups + 255) / 256;
  const uint32_t pf_magic = (uint32_t)((0x100000000ull + per_frame - 1) / per_frame);
  // vb / per_frame by multiply-high with ceil(2^32 / per_frame) is exact while (items of the launch) * per_frame < 2^32; a launch beyond that
  // (4096^2 planes from batch ~500 on) goes out in slices of whole frames that each stay inside the bound
  const uint32_t max_frames = (uint32_t)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1, 0xffffffffull / ((uint64_t)per_frame * per_frame)));
  for (int f0 = 0; f0 < batch; f0 += (int)max_frames) {
    const int nb = std::min<int>(batch - f0, (int)max_frames);
    const uint32_t per_xcd = (per_frame * (uint32_t)nb + 7) / 8;
    Level0View v = l0;
    if (v.vbase) v.vbase += (int64_t)f0 * v.frame_stride;
    hipLaunchKernelGGL(k_resize_level, dim3(8 * per_xcd), block, 0, s, d_pyr + (int64_t)f0 * pyr_block, pyr_block, src.plane_off, src.pitch, src.w, dst.plane_off, dst.pitch,
                       dst.ph, dst.w, fast_ok, d_ctab, d_rtab, magic, v, src.h, per_frame, pf_magic, per_xcd, nb, nwx, wx0, rg0, row_end);
  }
}


// =====================================================================================================================
// k_pyr_tiles: a GROUP of consecutive pyramid levels in one launch (plan: pyr_tiles.hpp).  Workgroup = (frame, tile); it walks the group's
// levels in turn: level l is computed from the tile of level l - 1 the workgroup left in LDS (the group's first level: from memory) over
// its own cell -- stored to the padded plane -- and the halo its cell of level l + 1 reads -- stored to LDS only; one workgroup barrier per
// level.  Work item = 4 rows x one dword column, exactly k_resize_level's arithmetic and tables (the 12-byte tap window, two v_perm per
// source row, 11-bit weights x 16), so the planes are the per-level launches' planes byte for byte.
struct PyrTilesArgs {
  uint8_t* pyr;
  int64_t pyr_block;
  const PyrTileLevel* plan;  // [tile][nlevels]
  const ResizeCol* ctab;
  const ResizeRow* rtab;
  Level0View l0;             // level 1 reads the caller's image in place (vbase != NULL)
  int nlevels, first, last, ntiles, batch;
  uint32_t ntiles_magic, per_xcd;
  // the group's source (level first - 1 in memory) and per-level constants of the handle's geometry
  int64_t src_plane_off;
  int src_pitch, src_w, src_h;
  int64_t plane_off[kMaxLevels];
  int pitch[kMaxLevels], xtab_off[kMaxLevels], ytab_off[kMaxLevels];
};

constexpr int kPyrTilesMaxGroup = 8;  // levels above the first whose tables one pass of the staging code covers (deeper groups loop)

// One level of a workgroup's walk.  FROM_LDS: the source is the LDS tile of level l - 1 (SP) and the coefficient tables are the copies
// staged in LDS; otherwise (the group's first level) source and tables are read from memory.  ROWS = output rows per work item: 4 (the
// column entries are fetched once per four rows: the throughput shape) or 1 (four times as many items: a handful of frames spread over
// 1024-thread workgroups, where a level is a latency chain and not a matter of instruction counts).
template <bool FROM_LDS, int ROWS>
__device__ __forceinline__ void pyr_tile_level(const PyrTilesArgs& A, const PyrTileLevel& T, const PyrTileLevel& SP, int l, int f, uint8_t* lds) {
  static_assert(ROWS == 1 || ROWS == RZ_ROWS, "items are single rows or row groups of the row table");
  const ResizeCol* __restrict__ ct = A.ctab + A.xtab_off[l];
  const ResizeRow* __restrict__ rt = A.rtab + A.ytab_off[l];
  const uint4* __restrict__ tabs = reinterpret_cast<const uint4*>(lds + T.tab_off);
  const int ncw = (int)T.ncw;
  const int nitems = ncw * (int)T.nrg * (RZ_ROWS / ROWS);
  // source: the LDS tile of level l - 1 (byte (0, 0) = ROI pixel (lx0, ly0)), or the ROI of level l - 1 in memory
  const bool ip = !FROM_LDS && l == 1 && A.l0.vbase != nullptr;
  const int src_pitch = FROM_LDS ? (int)SP.lw + kPyrTileSlack : (ip ? A.l0.pitch : A.src_pitch);
  const uint8_t* S = nullptr;
  if (!FROM_LDS) S = ip ? A.l0.vbase + f * A.l0.frame_stride + (int64_t)kPad * src_pitch + kPad : A.pyr + f * A.pyr_block + A.src_plane_off + (int64_t)kPad * src_pitch + kPad;
  const uint8_t* SLDS = lds + SP.lds_off - ((int)SP.ly0 * src_pitch + (int)SP.lx0);  // address of ROI pixel (0, 0) if the tile began there
  const int sw = A.src_w, sh = A.src_h;
  const int64_t s_end = ip ? (int64_t)(sh - 1) * src_pitch + sw : INT64_MAX;
  uint8_t* dplane = A.pyr + f * A.pyr_block + A.plane_off[l];
  const int dpitch = A.pitch[l];
  const int tpitch = (int)T.lw + kPyrTileSlack;
  uint8_t* TL = lds + T.lds_off;
  for (int it = (int)threadIdx.x; it < nitems; it += (int)blockDim.x) {
    const int rl = (int)__umulhi((uint32_t)it, T.ncw_magic);  // row group (ROWS = 4) or row (ROWS = 1) of the computed region
    const int cwl = it - rl * ncw;
    const int wx = (int)T.cx0w + cwl;
    const int py0 = (int)T.cy0 + rl * ROWS;
    uint4 c01, c23;
    uint32_t tw[2 * ROWS];
    if (FROM_LDS) {
      c01 = tabs[2 * cwl], c23 = tabs[2 * cwl + 1];
      if (ROWS == RZ_ROWS) {
        const uint4 t01 = tabs[2 * ncw + 2 * rl], t23 = tabs[2 * ncw + 2 * rl + 1];
        const uint32_t q[8] = {t01.x, t01.y, t01.z, t01.w, t23.x, t23.y, t23.z, t23.w};
#pragma unroll
        for (int j = 0; j < 2 * ROWS; ++j) tw[j] = q[j];
      } else {
        const uint2 t0 = reinterpret_cast<const uint2*>(tabs + 2 * ncw)[rl];
        tw[0] = t0.x, tw[1] = t0.y;
      }
    } else {
      c01 = reinterpret_cast<const uint4*>(ct)[wx * 2], c23 = reinterpret_cast<const uint4*>(ct)[wx * 2 + 1];
      if (ROWS == RZ_ROWS) {
        const int rg = py0 >> 2;
        const uint4 t01 = reinterpret_cast<const uint4*>(rt)[rg * 2], t23 = reinterpret_cast<const uint4*>(rt)[rg * 2 + 1];
        const uint32_t q[8] = {t01.x, t01.y, t01.z, t01.w, t23.x, t23.y, t23.z, t23.w};
#pragma unroll
        for (int j = 0; j < 2 * ROWS; ++j) tw[j] = q[j];
      } else {
        const uint2 t0 = reinterpret_cast<const uint2*>(rt)[py0];
        tw[0] = t0.x, tw[1] = t0.y;
      }
    }
    const uint32_t cw[8] = {c01.x, c01.y, c01.z, c01.w, c23.x, c23.y, c23.z, c23.w};
    uint32_t a0[4], a1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = cw[2 * i] >> 16, a1[i] = cw[2 * i + 1] & 0xffffu;
    const uint32_t base = cw[1] >> 16;
    const uint32_t sel = (cw[3] >> 16) | (cw[5] & 0xffff0000u);
    int sy0[ROWS], sy1[ROWS];
    uint32_t b0s[ROWS], b1s[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      sy0[j] = (int)(tw[2 * j] & 0xffffu), sy1[j] = (int)(tw[2 * j] >> 16);
      b0s[j] = (tw[2 * j + 1] & 0xffffu) << 8, b1s[j] = (tw[2 * j + 1] >> 16) << 8;
    }
    uint32_t u[ROWS][3], w[ROWS][3];
    if (FROM_LDS) {
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        const uint32_t* p0 = reinterpret_cast<const uint32_t*>(SLDS + sy0[j] * src_pitch + (int)base);
        const uint32_t* p1 = reinterpret_cast<const uint32_t*>(SLDS + sy1[j] * src_pitch + (int)base);
        u[j][0] = p0[0], u[j][1] = p0[1], u[j][2] = p0[2];
        w[j][0] = p1[0], w[j][1] = p1[1], w[j][2] = p1[2];
      }
    } else {
      // (in place, only an item that reaches the frame's last source row has to look at the frame's end: k_resize_level)
      int sy_max = sy1[0];
#pragma unroll
      for (int j = 1; j < ROWS; ++j) sy_max = max(sy_max, sy1[j]);
      const bool guard = ip && sy_max >= sh - 1;
      if (!guard) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
          const uint32_t* p0 = reinterpret_cast<const uint32_t*>(S + (int64_t)sy0[j] * src_pitch + base);
          const uint32_t* p1 = reinterpret_cast<const uint32_t*>(S + (int64_t)sy1[j] * src_pitch + base);
          u[j][0] = p0[0], u[j][1] = p0[1], u[j][2] = p0[2];
          w[j][0] = p1[0], w[j][1] = p1[1], w[j][2] = p1[2];
        }
      } else {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
          const int64_t o0 = (int64_t)sy0[j] * src_pitch + base, o1 = (int64_t)sy1[j] * src_pitch + base;
          const uint32_t* p0 = reinterpret_cast<const uint32_t*>(S + o0);
          const uint32_t* p1 = reinterpret_cast<const uint32_t*>(S + o1);
          u[j][0] = p0[0], u[j][1] = o0 + 4 < s_end ? p0[1] : 0u, u[j][2] = o0 + 8 < s_end ? p0[2] : 0u;
          w[j][0] = p1[0], w[j][1] = o1 + 4 < s_end ? p1[1] : 0u, w[j][2] = o1 + 8 < s_end ? p1[2] : 0u;
        }
      }
    }
    const bool own_x = wx >= (int)T.ox0w && wx < (int)T.ox1w;
    const int tx = wx * 4 - kPad - (int)T.lx0;  // byte column in this level's LDS tile
    const bool tile_x = tx >= 0 && tx < (int)T.lw;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const uint32_t L0 = __builtin_amdgcn_perm(u[j][1], u[j][0], sel);
      const uint32_t R0 = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(u[j][2], u[j][1], 1), __builtin_amdgcn_alignbyte(u[j][1], u[j][0], 1), sel);
      const uint32_t L1 = __builtin_amdgcn_perm(w[j][1], w[j][0], sel);
      const uint32_t R1 = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(w[j][2], w[j][1], 1), __builtin_amdgcn_alignbyte(w[j][1], w[j][0], 1), sel);
      uint32_t v = vrow(hrow<0>(L0, R0, a0[0], a1[0]), hrow<0>(L1, R1, a0[0], a1[0]), b0s[j], b1s[j]);
      v |= vrow(hrow<1>(L0, R0, a0[1], a1[1]), hrow<1>(L1, R1, a0[1], a1[1]), b0s[j], b1s[j]) << 8;
      v |= vrow(hrow<2>(L0, R0, a0[2], a1[2]), hrow<2>(L1, R1, a0[2], a1[2]), b0s[j], b1s[j]) << 16;
      v |= vrow(hrow<3>(L0, R0, a0[3], a1[3]), hrow<3>(L1, R1, a0[3], a1[3]), b0s[j], b1s[j]) << 24;
      const int py = py0 + j;
      if (own_x && py >= (int)T.oy0 && py < (int)T.oy1) *reinterpret_cast<uint32_t*>(dplane + (int64_t)py * dpitch + wx * 4) = v;
      const int ty = py - kPad - (int)T.ly0;
      if (tile_x && ty >= 0 && ty < (int)T.lrows) *reinterpret_cast<uint32_t*>(TL + ty * tpitch + tx) = v;
    }
  }
}

template <int THREADS, int ROWS>
__global__ __launch_bounds__(THREADS) void k_pyr_tiles(PyrTilesArgs A) {
  extern __shared__ __align__(16) uint8_t pyr_lds[];
  // workgroup b runs on XCD b & 7; an XCD walks whole frames (a frame's tiles meet in one L2: the lines two tiles share are merged there)
  const uint32_t vb = (blockIdx.x & 7u) * A.per_xcd + (blockIdx.x >> 3);
  if (vb >= (uint32_t)A.ntiles * (uint32_t)A.batch) return;
  const int f = A.ntiles > 1 ? (int)__umulhi(vb, A.ntiles_magic) : (int)vb;
  const int t = (int)vb - f * A.ntiles;
  const PyrTileLevel* __restrict__ TP = A.plan + (size_t)t * A.nlevels;
  // The coefficient tables of the levels above the first go to LDS now -- every load in flight at once, beside the first level's own
  // fetches -- so that from the second level on nothing waits for memory: a level is LDS reads, arithmetic, stores and a barrier.
  // Per level (ncw + nrg) x 2 uint4: the column entries of the dword columns it computes, the row entries of its row groups.
  {
    uint4 v[kPyrTilesMaxGroup];
#pragma unroll
    for (int k = 0; k < kPyrTilesMaxGroup; ++k) {
      const int l = A.first + 1 + k;
      if (l <= A.last) {
        const PyrTileLevel& T = TP[l];
        const int nc = 2 * (int)T.ncw, n = nc + 2 * (int)T.nrg, i = (int)threadIdx.x;
        if (i < n)
          v[k] = i < nc ? reinterpret_cast<const uint4*>(A.ctab + A.xtab_off[l])[2 * (int)T.cx0w + i]
                        : reinterpret_cast<const uint4*>(A.rtab + A.ytab_off[l])[2 * ((int)T.cy0 >> 2) + (i - nc)];
      }
    }
    pyr_tile_level<false, ROWS>(A, TP[A.first], TP[A.first], A.first, f, pyr_lds);
#pragma unroll
    for (int k = 0; k < kPyrTilesMaxGroup; ++k) {
      const int l = A.first + 1 + k;
      if (l <= A.last) {
        const PyrTileLevel& T = TP[l];
        const int nc = 2 * (int)T.ncw, n = nc + 2 * (int)T.nrg;
        uint4* dst = reinterpret_cast<uint4*>(pyr_lds + T.tab_off);
        if ((int)threadIdx.x < n) dst[threadIdx.x] = v[k];
        for (int i = (int)threadIdx.x + THREADS; i < n; i += THREADS)  // (tiles wider than a workgroup has threads)
          dst[i] = i < nc ? reinterpret_cast<const uint4*>(A.ctab + A.xtab_off[l])[2 * (int)T.cx0w + i]
                          : reinterpret_cast<const uint4*>(A.rtab + A.ytab_off[l])[2 * ((int)T.cy0 >> 2) + (i - nc)];
      }
    }
    for (int l = A.first + 1 + kPyrTilesMaxGroup; l <= A.last; ++l) {  // (groups deeper than one pass covers)
      const PyrTileLevel& T = TP[l];
      const int nc = 2 * (int)T.ncw, n = nc + 2 * (int)T.nrg;
      uint4* dst = reinterpret_cast<uint4*>(pyr_lds + T.tab_off);
      for (int i = (int)threadIdx.x; i < n; i += THREADS)
        dst[i] = i < nc ? reinterpret_cast<const uint4*>(A.ctab + A.xtab_off[l])[2 * (int)T.cx0w + i]
                        : reinterpret_cast<const uint4*>(A.rtab + A.ytab_off[l])[2 * ((int)T.cy0 >> 2) + (i - nc)];
    }
  }
  for (int l = A.first + 1; l <= A.last; ++l) {
    __syncthreads();
    pyr_tile_level<true, ROWS>(A, TP[l], TP[l - 1], l, f, pyr_lds);
  }
}

// the tiles of a plan may take up to kPyrTilesMaxLds bytes of LDS: the kernels' limit is raised to it once per device (hipFuncSetAttribute
// sets the limit on the current device).  A device or runtime that refuses leaves the kernels at the 64 KB every launch may ask for: the
// tile pyramid is an optional launch shape, so *max_lds reports what plans may use and build_tile_set falls back to the per-level launches
// for plans that do not fit -- the extractor itself never fails over it.
int prepare_pyr_tiles(uint32_t* max_lds) {
  static std::mutex mu;
  static uint32_t limit[64] = {0};
  int dev = 0;
  UVO_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  uint32_t got = dev >= 0 && dev < 64 ? limit[dev] : 0;
  if (!got) {
    const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_tiles<256, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kPyrTilesMaxLds) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_tiles<1024, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kPyrTilesMaxLds) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_tiles<1024, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kPyrTilesMaxLds) == hipSuccess;
    if (!ok) (void)hipGetLastError();  // (not an error of the extractor: the plans are just held to the default limit)
    got = ok ? (uint32_t)kPyrTilesMaxLds : 64u * 1024u;
    if (dev >= 0 && dev < 64) limit[dev] = got;
  }
  if (max_lds) *max_lds = got;
  return UVO_OK;
}

int launch_pyr_tiles(hipStream_t s, uint8_t* d_pyr, int64_t pyr_block, const PyrTileLevel* d_plan, const Geom& g, const ResizeCol* d_ctab, const ResizeRow* d_rtab,
                     Level0View l0, int first, int last, int ntiles, uint32_t lds_bytes, int threads, int rows, int batch) {
  PyrTilesArgs A;
  memset(&A, 0, sizeof(A));
  A.pyr = d_pyr, A.pyr_block = pyr_block, A.plan = d_plan, A.ctab = d_ctab, A.rtab = d_rtab, A.l0 = first == 1 ? l0 : Level0View{nullptr, 0, 0, 0};
  A.nlevels = g.nlevels, A.first = first, A.last = last, A.ntiles = ntiles;
  A.src_plane_off = g.lv[first - 1].plane_off, A.src_pitch = g.lv[first - 1].pitch, A.src_w = g.lv[first - 1].w, A.src_h = g.lv[first - 1].h;
  for (int l = 0; l < g.nlevels; ++l) A.plane_off[l] = g.lv[l].plane_off, A.pitch[l] = g.lv[l].pitch, A.xtab_off[l] = g.lv[l].xtab_off, A.ytab_off[l] = g.lv[l].ytab_off;
  A.ntiles_magic = ntiles > 1 ? (uint32_t)((0x100000000ull + (uint32_t)ntiles - 1) / (uint32_t)ntiles) : 0u;  // (one tile per frame: the kernel takes f = vb)
  // (frame index by multiply-high: exact while items * ntiles < 2^32; larger launches go out in slices of whole frames)
  const int max_frames = (int)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1, 0xffffffffull / ((uint64_t)ntiles * ntiles)));
  for (int f0 = 0; f0 < batch; f0 += max_frames) {
    const int nb = std::min(batch - f0, max_frames);
    A.batch = nb, A.per_xcd = ((uint32_t)ntiles * (uint32_t)nb + 7) / 8;
    PyrTilesArgs B = A;
    B.pyr = d_pyr + (int64_t)f0 * pyr_block;
    if (B.l0.vbase) B.l0.vbase += (int64_t)f0 * B.l0.frame_stride;
    if (threads >= 1024 && rows == 1) hipLaunchKernelGGL((k_pyr_tiles<1024, 1>), dim3(8 * B.per_xcd), dim3(1024), lds_bytes, s, B);
    else if (threads >= 1024) hipLaunchKernelGGL((k_pyr_tiles<1024, 4>), dim3(8 * B.per_xcd), dim3(1024), lds_bytes, s, B);
    else hipLaunchKernelGGL((k_pyr_tiles<256, 4>), dim3(8 * B.per_xcd), dim3(256), lds_bytes, s, B);
  }
  return 0;
}

}  // namespace uvo
