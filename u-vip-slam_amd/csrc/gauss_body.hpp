// The blur's device code (see gauss.hip for the description): shared by k_gauss7 and by k_octree_gauss (octree.hip), whose workgroups
// that are not quad-tree problems run it.
#pragma once
#include "common.hpp"
#include "fast_geom.hpp"
#include <type_traits>

namespace uvo {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk(uint32_t x) { return __builtin_bit_cast(u16x2, x); }

// Row pass for the lane's 4 pixels with the packed-byte dot product (v_dot4_u32_u8).  Pixel k needs window bytes k+1 .. k+7 of [L C R]:
// instead of aligning the data to the taps (six v_alignbyte + eight dot products) the TAPS are aligned to the data -- every pixel takes the
// words L, C, R as they are, against tap vectors shifted to its place (wave-uniform constants: scalar operands) -- ten dot products, no byte
// shuffles.  Exact integer arithmetic.
//   pixel 0: L.(0,t0,t1,t2) + C.(t3,t2,t1,t0)          pixel 1: L.(0,0,t0,t1) + C.(t2,t3,t2,t1) + R.(t0,0,0,0)
//   pixel 2: L.(0,0,0,t0) + C.(t1,t2,t3,t2) + R.(t1,t0,0,0)          pixel 3: C.(t0,t1,t2,t3) + R.(t2,t1,t0,0)
struct GaussRowTaps {
  uint32_t l0, c0, l1, c1, r1, l2, c2, r2, c3, r3;
};
__device__ __forceinline__ GaussRowTaps gauss_row_taps(int4 t) {
  auto pk4 = [](int a, int b, int c, int d) { return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24); };
  GaussRowTaps T;
  T.l0 = pk4(0, t.x, t.y, t.z), T.c0 = pk4(t.w, t.z, t.y, t.x);
  T.l1 = pk4(0, 0, t.x, t.y), T.c1 = pk4(t.z, t.w, t.z, t.y), T.r1 = pk4(t.x, 0, 0, 0);
  T.l2 = pk4(0, 0, 0, t.x), T.c2 = pk4(t.y, t.z, t.w, t.z), T.r2 = pk4(t.y, t.x, 0, 0);
  T.c3 = pk4(t.x, t.y, t.z, t.w), T.r3 = pk4(t.z, t.y, t.x, 0);
  return T;
}
__device__ __forceinline__ void gauss_row_pass(uint32_t L, uint32_t C, uint32_t R, const GaussRowTaps& T, float* h) {
  const uint32_t s0 = __builtin_amdgcn_udot4(C, T.c0, __builtin_amdgcn_udot4(L, T.l0, 0u, false), false);
  const uint32_t s1 = __builtin_amdgcn_udot4(R, T.r1, __builtin_amdgcn_udot4(C, T.c1, __builtin_amdgcn_udot4(L, T.l1, 0u, false), false), false);
  const uint32_t s2 = __builtin_amdgcn_udot4(R, T.r2, __builtin_amdgcn_udot4(C, T.c2, __builtin_amdgcn_udot4(L, T.l2, 0u, false), false), false);
  const uint32_t s3 = __builtin_amdgcn_udot4(R, T.r3, __builtin_amdgcn_udot4(C, T.c3, 0u, false), false);
  h[0] = (float)s0, h[1] = (float)s1, h[2] = (float)s2, h[3] = (float)s3;  // < 2^16: exact
}

// the strip plans of a geometry's levels (region = ROI + the 4-pixel ring: (w + 8) x (h + 8)), made on the host once per launch: a
// wavefront that finds its level by planning every level in front of it spends three integer divisions per level on it
struct GaussPlans {
  StripPlan p[kMaxLevels];
};
int gauss7_rows_per_seg(int batch);
int gauss7_blocks_per_frame(const Geom& g, int rows_per_seg);
GaussPlans gauss7_plans(const Geom& g, int rows_per_seg);
constexpr int GS_TILES = 16;  // tiles a wavefront collects per 8-row group: 15 of a full strip, 2 x 7 / 4 x 3 of the narrow ones
// SSE2: the rounding contract of an x86-64 OpenCV build (UVO_TUNE_BLUR_ROUNDING, the default): SymmColumnVec_32s8u's vector body -- image
// columns 0 .. (w & ~3) - 1 -- converts the exact fp32 column sum with cvtps2dq + two saturating packs, i.e. round to nearest, an exact .5 to
// the EVEN neighbour, clamp to 255: exactly what v_cvt_pk_u8_f32 does under the wavefront's default rounding mode (tools/ubench/
// cvt_pk_u8_round.hip), so the conversion IS the contract.  The last w % 4 columns (the scalar tail) round .5 up: floor(sum + .5), in the few
// wavefronts that hold such columns.  A lane's four pixels are an aligned group of four image columns, so a lane is wholly one or the
// other.  !SSE2 (a build without SIMD): .5 up on every column -- the wavefront runs in round-toward-zero mode and the conversion is the floor.
constexpr int GS_TILE_DW = GS_TILES * 32 + 96;   // LDS dwords per wavefront: the tiles + a dword per lane (and row phase) for the lanes that collect nothing
constexpr int GS_LDS_BYTES = 4 * GS_TILE_DW * 4;  // per 4-wavefront workgroup
// The body of a k_gauss7 workgroup: `block` of `blocks_x * batch` (frame-major in an XCD-contiguous order), LDS = GS_LDS_BYTES at s_tile.
// Also run by the workgroups of k_octree_gauss (octree.hip) that are not quad-tree problems.
template <bool SSE2>
__device__ __forceinline__ void gauss7_body(int block, int blocks_x, int batch, uint32_t (*s_tile)[GS_TILE_DW], const uint8_t* __restrict__ pyr,
                                            uint8_t* __restrict__ blur, int64_t pyr_block, const LevelGeom* __restrict__ lv, int nlevels, int4 taps, int rows_per_seg,
                                            Level0View l0, const GaussPlans& plans) {
  // work item (one per wavefront) -> (level, strip, segment group); narrow remainder strips hold 2 or 4 row segments side by side
  // (fast_strip_plan), so a level costs about as many wavefront-rows as its width needs
  const int vb = xcd_contiguous(block, blocks_x * batch);
  const int vbx = vb % blocks_x, f = vb / blocks_x;  // an XCD walks whole frames, item after item
  int item = vbx * 4 + wave_in_block();
  const int lane = threadIdx.x & 63;
  int level = 0;
  for (;; ++level) {
    if (item < plans.p[level].items) break;
    item -= plans.p[level].items;
    if (level == nlevels - 1) return;
  }
  const StripPlan plan = plans.p[level];
  const LevelGeom g = lv[level];
  int strip_x, seg, nsub;
  fast_strip_item(plan, item, strip_x, seg, nsub);
  // The walk, in two instantiations: over a padded plane, or (IP) over level 0 read in place -- the caller's image, whose 16-pixel
  // REFLECT_101 border (the blur reaches 3 pixels of it, and copies 4 into the blurred plane's ring) is produced by reflecting the row
  // index and, per lane, the column run: a lane's four pixels are an aligned group of four image columns (the width is a multiple of 4 in
  // this mode), so a group is wholly inside the image or wholly a reversed run of it -- one load and one byte permute either way.
  auto walk = [&](auto ip_tag, auto tail_tag) {
  constexpr bool IP = decltype(ip_tag)::value;
  constexpr bool TAIL = decltype(tail_tag)::value;  // SSE2 contract: the wavefront holds output columns of the scalar tail (decided once, below: no test in the row loop)
  const uint8_t* src = IP ? l0.vbase + f * l0.frame_stride + (int64_t)kPad * l0.pitch + kPad : pyr + f * pyr_block + g.plane_off;  // IP: the image's origin
  const int spitch = IP ? l0.pitch : g.pitch;
  uint8_t* dst = blur + f * pyr_block + g.plane_off;

  const int lps = 64 / nsub, sub = (lane * nsub) >> 6, ls = lane - sub * lps;
  // padded-plane column of this lane's dword; the first lane of a sub-strip is its left halo.  Region = padded cols [12, w+20).
  const int X = 8 + strip_x + ls * 4;
  int Xc = X > g.pitch - 4 ? g.pitch - 4 : X;  // clamp loads into the row (only halo / out-of-region lanes)
  uint32_t csel = 0x03020100u;                 // IP: identity, or byte reversal for a reflected run
  if (IP) {
    const int x = X - kPad;                    // image column of the lane's first pixel (a multiple of 4)
    int run = x;
    if (x < 0) run = -x - 3, csel = 0x00010203u;                           // columns x .. x+3 = image columns -x .. -x-3
    else if (x > g.w - 4) run = 2 * (g.w - 1) - x - 3, csel = 0x00010203u;  // columns x .. x+3 = image columns 2(w-1)-x .. 2(w-1)-x-3
    Xc = run < 0 ? 0 : (run > g.w - 4 ? g.w - 4 : run);                    // (lanes farther out than the border are never used)
  }
  // padded-plane rows: region rows [12, h+20); the lane's segment rows [py0l, py1l)
  const int py0 = 12 + seg * rows_per_seg;
  const int py0l = py0 + sub * rows_per_seg;
  const int py1l = min(py0l + rows_per_seg, g.h + 20);
  const bool lane_out = ls >= 1 && ls <= lps - 2 && X >= 12 && X < g.w + 20 && py1l > py0l;
  const int tile_col = (X >> 4) * 128 + (X & 15);   // the lane's place inside a row of tiles
  const int64_t tile_row_bytes = (int64_t)g.pitch * 8;  // (pitch / 16) tiles of 128 bytes
  // Output goes to the plane in whole cache lines: the tiles a sub-strip's output lanes cover completely (columns
  // [16 t_first, 16 (t_last + 1)) of its output range) are collected in LDS -- eight rows of the walk fill them -- and written out as
  // 16 bytes per lane, eight lanes per 128-byte line; the few columns left and right of them are stored directly, a dword per row.
  const int x_lo = max(12, 8 + strip_x + 4), x_hi = min(g.w + 20, 8 + strip_x + 4 * (lps - 1));  // output columns of the sub-strip
  const int t_first = (x_lo + 15) >> 4, nts = max((x_hi >> 4) - t_first, 0);                    // its full tiles: t_first .. t_first + nts - 1
  const int ts_shift = nsub == 1 ? 4 : (nsub == 2 ? 3 : 2), TS = 1 << ts_shift;                 // LDS tile slots per sub-strip (16 / nsub >= nts)
  const bool via_lds = lane_out && (X >> 4) >= t_first && (X >> 4) < t_first + nts;
  uint32_t* stile = s_tile[wave_in_block()];
  // + (py & 7) * 4: dword of the lane's pixels in the tile.  Lanes that collect nothing write their own dword behind the tiles -- the
  // LDS write of a row is then unconditional (no exec mask to set up and restore: scalar instructions are not free)
  const int lds_slot = via_lds ? ((sub * TS + ((X >> 4) - t_first)) << 5) + ((X & 15) >> 2) : GS_TILES * 32 + lane;
  // The write-out of a group: task t = lane + 64 it (it = 0, 1) is row t & 7 of tile slot t >> 3; everything about a task that does not
  // depend on the group is worked out here, once.
  // (row addresses: a sub-strip's rows lie s * rows_per_seg below sub-strip 0's, a whole number of tile rows, so every address is
  // "tile row of sub-strip 0" -- wave-uniform, scalar arithmetic -- plus a per-lane constant)
  const int64_t seg_tile_rows = (int64_t)(rows_per_seg >> 3) * tile_row_bytes;
  const int64_t lane_goff = (int64_t)sub * seg_tile_rows + tile_col;  // the lane's own dword relative to sub-strip 0's tile row
  int f_lds[2], f_p0[2], f_p1[2], f_rowoff[2];
  int64_t f_goff[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = lane + 64 * it, row = t & 7, slot = t >> 3, s_ = slot >> ts_shift, ti = slot & (TS - 1);
    const int p0 = py0 + s_ * rows_per_seg;
    f_lds[it] = (slot << 5) + (row << 2);
    f_goff[it] = (int64_t)s_ * seg_tile_rows + (t_first + ti) * 128 + (row << 4);
    f_rowoff[it] = s_ * rows_per_seg + row;
    f_p0[it] = p0, f_p1[it] = (s_ < nsub && ti < nts) ? min(p0 + rows_per_seg, g.h + 20) : p0;  // empty range: no such tile
  }
  // writes rows 0 .. r_hi of the 8-row group that starts at padded row pyg (of sub-strip 0) of every collected tile to the plane
  auto flush_tiles = [&](int pyg, int r_hi, int64_t group_base) {  // group_base = byte offset of the group's tile row (sub-strip 0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int prow = pyg + f_rowoff[it];  // the row of the task's own sub-strip in the plane
      if ((lane & 7) <= r_hi && prow >= f_p0[it] && prow < f_p1[it]) {
        const uint4 v = *reinterpret_cast<const uint4*>(stile + f_lds[it]);
        *reinterpret_cast<uint4*>(dst + (group_base + f_goff[it])) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  };

  const bool tail_lane = SSE2 && (g.w & 3) != 0 && (X - kPad) == (g.w & ~3);  // the lane's image columns are the scalar tail (w % 4 of them inside the image)
  uint32_t inmask = 0;  // bytes of the lane's dword that lie inside the image columns
#pragma unroll
  for (int k = 0; k < 4; ++k) inmask |= (X + k >= kPad && X + k < g.w + kPad) ? 0xffu << (8 * k) : 0u;
  const int nsrc = min(py0 + rows_per_seg, g.h + 20) - py0 + 6;  // source rows py0-3 .. py1+2 of sub-strip 0 (the longest)
  // per lane, in loop rows: output row py = py0l + j - 6 lies inside the image rows for j in [j_in0, j_in0 + n_in), and the lane stores
  // it directly (an output lane outside the collected tiles) for j < j_edge1
  const uint32_t j_in0 = (uint32_t)(kPad - py0l + 6), n_in = (uint32_t)g.h;
  const uint32_t j_edge1 = (lane_out && !via_lds) ? (uint32_t)max(py1l - py0l + 6, 0) : 0u;

  const GaussRowTaps RT = gauss_row_taps(taps);
  // column taps scaled by 2^-16 (exact): the column sum comes out as sum / 65536 -- every partial sum an exact multiple of 2^-16 below 2^8
  // (a larger one belongs to a result that saturates anyway), so no operation of the pass rounds, whatever the mode
  const float c0 = (float)taps.x * (1.0f / 65536.0f), c1 = (float)taps.y * (1.0f / 65536.0f), c2 = (float)taps.z * (1.0f / 65536.0f),
              c3 = (float)taps.w * (1.0f / 65536.0f);
  if (!SSE2) __builtin_amdgcn_s_setreg(0x801, 3);  // MODE.fp_round[1:0] (fp32) = toward zero: the float -> byte conversion of sum + .5 is the floor
  const int lane_up = (lane > 0 ? lane - 1 : lane) * 4, lane_down = (lane < 63 ? lane + 1 : lane) * 4;  // ds_bpermute byte addresses
  float hring[7][4];
  uint32_t cring[7];
  // Row loads are issued one unrolled block (7 rows) ahead of their use: a wavefront walks ~70 rows one after another, so
  // without the prefetch every row would expose a full memory round trip.
  auto load_row = [&](int j) -> uint32_t {
    int prow = py0l - 3 + j;
    if (IP) {
      int r = prow - kPad;                      // image row; REFLECT_101 above and below
      r = r < 0 ? -r : r;
      r = r > g.h - 1 ? 2 * (g.h - 1) - r : r;
      r = r < 0 ? 0 : r;                        // (rows past the border's reach are never used by a valid output)
      const uint32_t v = *reinterpret_cast<const uint32_t*>(src + (__umul24((uint32_t)r, (uint32_t)spitch) + (uint32_t)Xc));  // (see below)
      return __builtin_amdgcn_perm(v, v, csel);
    }
    prow = prow > g.ph - 1 ? g.ph - 1 : prow;  // rows past the plane are never used by a valid output
    // wave-uniform base + 32-bit lane offset by a 24-bit multiply (a plane is far below 4 GB; a 64-bit multiply-add issues at a quarter of the rate)
    return *reinterpret_cast<const uint32_t*>(src + (__umul24((uint32_t)prow, (uint32_t)spitch) + (uint32_t)Xc));
  };
  // byte offset of sub-strip 0's current output row inside its tile column: tile row * tile_row_bytes + (row % 8) * 16, advanced row by
  // row (one scalar add instead of a 64-bit multiply per row)
  int64_t rowb = (int64_t)(py0 >> 3) * tile_row_bytes + ((py0 & 7) << 4);
  int phase = (py0 - 6) & 7;  // (row of sub-strip 0) & 7 of loop row j = 0, advanced with it
  uint32_t cur[7], nxt[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) cur[u] = load_row(u);  // unconditional (the row index is clamped into the plane): with a branch around a load
                                                     // the compiler cannot count the loads in flight and waits for all of them
  for (int base = 0; base < nsrc; base += 7) {
#pragma unroll
    for (int u = 0; u < 7; ++u) nxt[u] = load_row(base + 7 + u);
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int j = base + u;
      if (j < nsrc) {
        const uint32_t C = cur[u];
        const uint32_t L = (uint32_t)__builtin_amdgcn_ds_bpermute(lane_up, (int)C);    // the neighbours' dwords: source lanes fixed for the
        const uint32_t R = (uint32_t)__builtin_amdgcn_ds_bpermute(lane_down, (int)C);  // whole strip (__shfl_up / _down recompute them per call)
        gauss_row_pass(L, C, R, RT, hring[u]);
        cring[u] = C;
        const int phase_now = phase;
        phase = (phase + 1) & 7;
        if (j >= 6) {
          // output row py = py0l + j - 6; its 7 source rows sit in ring slots (u+1)%7 .. (u+7)%7
          const float* r0 = hring[(u + 1) % 7];
          const float* r1 = hring[(u + 2) % 7];
          const float* r2 = hring[(u + 3) % 7];
          const float* r3 = hring[(u + 4) % 7];
          const float* r4 = hring[(u + 5) % 7];
          const float* r5 = hring[(u + 6) % 7];
          const float* r6 = hring[u];
          const uint32_t centre = cring[(u + 4) % 7];
          const bool row_in = (uint32_t)j - j_in0 < n_in;
          // column pass in fp32, one pixel per instruction (v_add_f32 / v_fma_f32); clamp to 255 and the byte insert are the conversion
          // itself.  (Round 5's packed form -- v_pk_add_f32 / v_pk_fma_f32 on pixel pairs -- issues fewer instructions and wins the
          // micro-benchmark, tools/ubench/pk_f32_rate.hip, but needs more registers: 0.238 - 0.246 against 0.229 - 0.231 ms per 257-frame
          // launch at the same occupancy, profiles/r06_blur_occupancy_ab.txt.)
          uint32_t blurred = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float z = __builtin_fmaf(c0, r0[k] + r6[k], __builtin_fmaf(c1, r1[k] + r5[k], __builtin_fmaf(c2, r2[k] + r4[k], __builtin_fmaf(c3, r3[k], SSE2 ? 0.0f : 0.5f))));
            if (SSE2 && TAIL) z = tail_lane ? __builtin_floorf(z + 0.5f) : z;  // the scalar tail's columns: floor(sum + .5) is a whole number, which the conversion leaves alone
            blurred = __builtin_amdgcn_cvt_pk_u8_f32(z, (uint32_t)k, blurred);
          }
          // pad ring and anything outside the image: the un-blurred centre pixel (byte mask per lane, rows uniform)
          const uint32_t m = row_in ? inmask : 0u;
          const uint32_t out = (blurred & m) | (centre & ~m);
          // tiled store: pixel (x, y) of the blurred plane lives in tile (y / 8, x / 16) -- 128 bytes, one cache line -- at byte
          // (y % 8) * 16 + x % 16 (a lane's four pixels never straddle a tile: X is a multiple of 4).  Four lanes fill a tile row,
          // eight consecutive rows of the walk complete the line in L2.
          // (py & 7 is the same for every sub-strip: their first rows differ by multiples of rows_per_seg, a multiple of 8)
          stile[lds_slot + (phase_now << 2)] = out;
          if ((uint32_t)j < j_edge1) *reinterpret_cast<uint32_t*>(dst + rowb + lane_goff) = out;
          if (phase_now == 7) {
            flush_tiles(py0 + j - 6 - 7, 7, rowb - 112);
            rowb += tile_row_bytes - 112;
          } else {
            rowb += 16;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) cur[u] = nxt[u];
  }
  {  // the rows of the last, incomplete group
    const int py_last = py0 + nsrc - 7;  // last output row of sub-strip 0
    if ((py_last & 7) != 7) flush_tiles(py_last & ~7, py_last & 7, (int64_t)(py_last >> 3) * tile_row_bytes);
  }
  };  // walk
  // SSE2: does this wavefront hold output columns of the scalar tail at all?  Only the last strip of a level whose width is no multiple of 4
  // (and never level 0 in place, whose width is one): those wavefronts take the instantiation with the fix-up in its column pass.
  const int x_last = 8 + strip_x + 4 * (64 / nsub - 1);  // first padded column behind the strip's lanes
  const bool tail_wave = SSE2 && (g.w & 3) != 0 && kPad + (g.w & ~3) >= 8 + strip_x && kPad + (g.w & ~3) < x_last;
  if (level == 0 && l0.vbase != nullptr)
    walk(std::true_type{}, std::false_type{});
  else if (tail_wave)
    walk(std::false_type{}, std::true_type{});
  else
    walk(std::false_type{}, std::false_type{});
}


}  // namespace uvo
