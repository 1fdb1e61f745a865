// NOT PART OF THE PRODUCT: the production-grade form of tools/ubench/mfma_blur.hip as it was measured in round 5 (profiles/r05_mfma_blur_probe.txt:
// bit-exact, 0.25 - 0.32 ms for the levels 1 - 7 of 257 frames against 0.17 for the vector-ALU blur of ALL levels).  Kept as the record of the attempt;
// it compiled as csrc/gauss_mfma_body.hpp behind a knob, with k_gauss_mfma / gauss_mfma_plan in csrc/gauss.hip.
// The 7 x 7 blur of a padded pyramid level on the int8 matrix cores (the other body is gauss_body.hpp: vector ALU, every level incl. level 0
// read in place).  Same planes, byte for byte: cv::GaussianBlur(7x7, sigma 2) of src/ORBextractor.cc:942 on the level's image, the 4-pixel
// ring around it copied from the un-blurred pad, output in the tiled layout k_describe reads (gauss.hip).
//
// The separable filter as matrix products, all integer and exact (prototype + measurements: tools/ubench/mfma_blur.hip):
//   row pass   : S[r][x] = sum_k tap[k] p[r][x + k - 3] = (32 rows x 32 input columns) x (32 x 32 banded Toeplitz) -- two
//                v_mfma_i32_32x32x32_i8 per 32 x 32 outputs (input columns [X - 16, X + 16) and [X + 16, X + 48)); an A operand is 16
//                consecutive pixels of ONE row per lane: a dwordx4 load.  Pixels biased by -128 (xor 0x80), the accumulator starts at
//                128 * sum(taps), so it holds S itself (16 bits).
//   column pass: out[y][x] = sum_k tap[k] S[y + k - 3][x].  S is split into hi / lo bytes (xor 0x80 each); the packed row-pass result
//                of a lane IS an A operand (m = column, k = the rows in the order the accumulator holds them) and B is the Toeplitz
//                in that row order; computed transposed, C'[x][y], so that a lane ends up with four consecutive x of one row per
//                register group.  Output rows are shifted by 16 against the input row blocks, so an output block takes two S blocks:
//                four MFMAs ((hi, lo) x (upper, lower)).
//   rounding   : sum / 65536 converted by v_cvt_pk_u8_f32 under the default mode = nearest even + clamp = the x86-64 contract;
//                floor(sum / 65536 + .5) for the scalar-tail columns and for the half-up contract (gauss_body.hpp).
// A wavefront owns a strip of 64 columns (two MFMA tiles) and walks down the row blocks; a block's 32 x 64 outputs go through 2 KB of
// LDS in the tiled layout and leave as whole 128-byte lines.  6 MFMAs + ~7 vector instructions per pixel instead of ~18.
#if 0  // a record, not a translation unit: the includes below are those of csrc/
#pragma once
#include "common.hpp"

namespace uvo {

typedef int gm_v4i __attribute__((ext_vector_type(4)));
typedef int gm_v16i __attribute__((ext_vector_type(16)));

constexpr int GM_STRIP = 64;       // output columns per wavefront
constexpr int GM_LDS_DW = 512;     // LDS dwords per wavefront: 32 rows x 64 bytes in tile order

struct GaussMfmaPlan {
  int first_level;                 // levels first_level .. nlevels - 1 take this body
  int nblk;                        // output row blocks (32 rows) a wavefront walks
  int items[kMaxLevels];           // wavefronts per frame and level = strips x segments (0 below first_level)
  int nstrips[kMaxLevels];
  int items_per_frame;
};

__device__ __forceinline__ int gm_tap(int4 t, int idx) {
  return idx == 0 || idx == 6 ? t.x : (idx == 1 || idx == 5 ? t.y : (idx == 2 || idx == 4 ? t.z : (idx == 3 ? t.w : 0)));
}
__device__ __forceinline__ int gm_pack4(int a, int b, int c, int d) {
  return (int)((uint32_t)(a & 255) | (uint32_t)(b & 255) << 8 | (uint32_t)(c & 255) << 16 | (uint32_t)(d & 255) << 24);
}

// item = the wavefront's work item inside frame f (level-major; strips fastest); stile = its GM_LDS_DW dwords of LDS
template <bool SSE2>
__device__ __forceinline__ void gauss_mfma_body(int item, int f, uint32_t* stile, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int64_t pyr_block,
                                                const LevelGeom* __restrict__ lv, int nlevels, int4 taps, const GaussMfmaPlan& plan) {
  int level = plan.first_level;
  for (;; ++level) {
    if (level >= nlevels) return;
    if (item < plan.items[level]) break;
    item -= plan.items[level];
  }
  const LevelGeom g = lv[level];
  const int nstr = plan.nstrips[level];
  const int seg = item / nstr, sx = item - seg * nstr;
  const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
  const int X0 = kPad + GM_STRIP * sx;                 // first output column (padded coordinates): the image starts at column 16
  const int nblocks = (g.h + 4 + 31) >> 5;             // output blocks of 32 rows from row 16 on cover the image and the ring below it
  const int ob0 = seg * plan.nblk, nob = min(plan.nblk, nblocks - ob0);
  if (nob <= 0) return;
  const uint8_t* src = pyr + f * pyr_block + g.plane_off;
  uint8_t* dst = blur + f * pyr_block + g.plane_off;
  const int pitch = g.pitch, tiles_x = pitch >> 4, tile_rows = (g.ph + 7) >> 3;
  const int sumt = 2 * (taps.x + taps.y + taps.z) + taps.w;

  // constant operands: the banded Toeplitz of the row pass (k = input column of the window) and of the column pass (k runs over the
  // rows in accumulator order: register i, byte j of lane half hh holds row 8 i + 4 hh + j of its block)
  gm_v4i B1, B2, T1, T2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q1[4], q2[4], r1[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * hh + 4 * i + j, ro = 8 * i + 4 * hh + j;
      q1[j] = gm_tap(taps, k - n - 13), q2[j] = gm_tap(taps, k - n + 19);
      r1[j] = gm_tap(taps, ro - n - 13), r2[j] = gm_tap(taps, ro - n + 19);
    }
    B1[i] = gm_pack4(q1[0], q1[1], q1[2], q1[3]), B2[i] = gm_pack4(q2[0], q2[1], q2[2], q2[3]);
    T1[i] = gm_pack4(r1[0], r1[1], r1[2], r1[3]), T2[i] = gm_pack4(r2[0], r2[1], r2[2], r2[3]);
  }
  gm_v16i zero16, rinit, linit;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0, rinit[i] = 128 * sumt, linit[i] = sumt * (32768 + 128);

  // per output dword (tile t, register group gg) of this lane: its padded column, which of its bytes are image columns, whether they are
  // the scalar tail of the x86-64 contract
  const bool edge_x = X0 + GM_STRIP > g.w + kPad;   // the strip reaches past the image's last column: ring / outside columns, maybe a tail
  uint32_t cmask[2][4];
  bool tail[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int pc = X0 + 32 * t + 8 * gg + 4 * hh;
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) m |= (pc + k < g.w + kPad) ? 0xffu << (8 * k) : 0u;
      cmask[t][gg] = m;
      tail[t][gg] = SSE2 && (g.w & 3) != 0 && pc - kPad == (g.w & ~3);
    }
  const bool any_tail = SSE2 && edge_x && (g.w & 3) != 0;

  // the lane's three 16-byte chunks of a row: columns X0 - 16 + 16 hh + 32 c (clamped into the row: what lies past the plane's pitch only
  // feeds columns outside the ring)
  int coff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) coff[c] = min(X0 - 16 + 16 * hh + 32 * c, pitch - 16);
  auto load_block = [&](int ib, gm_v4i* A) {
    const int r = min(32 * ib + n, g.ph - 1);  // rows past the plane feed rows below the ring only
    const uint8_t* p = src + (uint32_t)(r * pitch);
#pragma unroll
    for (int c = 0; c < 3; ++c) A[c] = *reinterpret_cast<const gm_v4i*>(p + coff[c]);
  };

  // ring rows above the image (padded rows 8 .. 15 = tile row 1; 12 .. 15 are read) and the ring columns left of it (tile column 0):
  // plain copies of the un-blurred pad, done by the wavefronts of the first segment / the first strip
  if (ob0 == 0 && lane < 32) {
    const int txl = lane >> 3, row = lane & 7;
    const int col = min(X0 + 16 * txl, pitch - 16);
    const uint4 v = *reinterpret_cast<const uint4*>(src + (uint32_t)((8 + row) * pitch + col));
    if ((X0 >> 4) + txl < tiles_x) *reinterpret_cast<uint4*>(dst + ((int64_t)(1 * tiles_x + (X0 >> 4) + txl) * 128 + row * 16)) = v;
  }
  if (sx == 0 && ob0 == 0 && lane >= 32 && lane < 40) {
    const int row = lane - 32;
    *reinterpret_cast<uint4*>(dst + ((int64_t)(1 * tiles_x) * 128 + row * 16)) = *reinterpret_cast<const uint4*>(src + (uint32_t)((8 + row) * pitch));
  }

  gm_v4i Anext[3], AhiP[2], AloP[2];
  load_block(ob0, Anext);
  for (int ib = ob0; ib <= ob0 + nob; ++ib) {
    gm_v4i A[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) A[c] = Anext[c];
    load_block(ib + 1, Anext);  // the next block's pixels are fetched while this one is multiplied
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) A[c][i] ^= (int)0x80808080;
    gm_v4i Ahi[2], Alo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      gm_v16i s = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t], B1, rinit, 0, 0, 0);
      s = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t + 1], B2, s, 0, 0, 0);
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const uint32_t a0 = (uint32_t)s[4 * gg], a1 = (uint32_t)s[4 * gg + 1], a2 = (uint32_t)s[4 * gg + 2], a3 = (uint32_t)s[4 * gg + 3];
        const uint32_t p01 = __builtin_amdgcn_perm(a1, a0, 0x05040100u);  // (lo0, hi0, lo1, hi1)
        const uint32_t p23 = __builtin_amdgcn_perm(a3, a2, 0x05040100u);
        Alo[t][gg] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
        Ahi[t][gg] = (int)(__builtin_amdgcn_perm(p23, p01, 0x07050301u) ^ 0x80808080u);
      }
    }
    if (ib > ob0) {
      // output block ob = ib - 1: rows 16 + 32 ob + n, from the S blocks ob (upper) and ob + 1 (lower)
      const int ob = ib - 1, py = kPad + 32 * ob + n;
      const bool row_in = py < g.h + kPad;
      const bool edge = edge_x || 32 * ob + 32 > g.h;   // (wave-uniform) some outputs of the block are ring / outside pixels: the un-blurred centre
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        gm_v16i accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(AhiP[t], T1, zero16, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ahi[t], T2, accH, 0, 0, 0);
        gm_v16i accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(AloP[t], T1, linit, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(Alo[t], T2, accL, 0, 0, 0);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          uint32_t out = 0;
          float z[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t S = ((uint32_t)accH[4 * gg + e] << 8) + (uint32_t)accL[4 * gg + e];
            z[e] = (float)S * (1.0f / 65536.0f);  // exact below 2^24 (a larger sum saturates whatever its rounding)
          }
          if (!SSE2 || (any_tail && tail[t][gg])) {
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = __builtin_floorf(z[e] + 0.5f);  // .5 up: a whole number, which the conversion leaves alone
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) out = __builtin_amdgcn_cvt_pk_u8_f32(z[e], (uint32_t)e, out);
          if (edge) {
            const int pc = X0 + 32 * t + 8 * gg + 4 * hh;
            const uint32_t centre = *reinterpret_cast<const uint32_t*>(src + (uint32_t)(min(py, g.ph - 1) * pitch + min(pc, pitch - 4)));
            const uint32_t m = row_in ? cmask[t][gg] : 0u;
            out = (out & m) | (centre & ~m);
          }
          // LDS in tile order: tile (n / 8, 2 t + gg / 2) of the block's 4 x 4, dword (n % 8) * 4 + 2 (gg % 2) + hh of its 32
          stile[(((n >> 3) * 4 + 2 * t + (gg >> 1)) << 5) + ((n & 7) << 2) + 2 * (gg & 1) + hh] = out;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // the block leaves as 16 whole tiles (cache lines): lane L carries 16 bytes of tile L / 8 (+ 8), row L % 8
      const int ty0 = 2 + 4 * ob, tx0 = X0 >> 4;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int ti = (lane >> 3) + 8 * it, ty = ty0 + (ti >> 2), tx = tx0 + (ti & 3);
        const uint4 v = *reinterpret_cast<const uint4*>(stile + (ti << 5) + ((lane & 7) << 2));
        if (ty < tile_rows && tx < tiles_x) *reinterpret_cast<uint4*>(dst + ((int64_t)(ty * tiles_x + tx) * 128 + ((lane & 7) << 4))) = v;
      }
      if (sx == 0 && lane < 32) {  // the ring columns left of the image: tile column 0 of the block's rows
        const int pyc = min(py, g.ph - 1);
        if ((py >> 3) < tile_rows)
          *reinterpret_cast<uint4*>(dst + ((int64_t)((py >> 3) * tiles_x) * 128 + ((py & 7) << 4))) = *reinterpret_cast<const uint4*>(src + (uint32_t)(pyc * pitch));
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) AhiP[t] = Ahi[t], AloP[t] = Alo[t];
  }
}

}  // namespace uvo

#endif
