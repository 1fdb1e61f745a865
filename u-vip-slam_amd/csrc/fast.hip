// Per-cell FAST-9/16 with in-cell 3x3 non-max suppression and the per-cell threshold fallback.
// Replaces the cell loop of ORBextractor::ComputeKeyPointsOctTree (src/ORBextractor.cc:773-812):
//   FAST(cellROI, kps, fastTh, true); if (kps.empty()) FAST(cellROI, kps, 7, true);
// (cv::FAST, TYPE_9_16: 9 contiguous circle pixels all > v+t or all < v-t; score = cornerScore<16> = the largest
// threshold that keeps the pixel a corner; NMS keeps strict 8-neighbour maxima, neighbours outside the ROI's
// 3-px-inset interior or that are not corners count as 0.)
//
// The cells' interiors tile the detection window [16, w-16) x [16, h-16) exactly once and the corner score does not
// depend on the threshold, so the work is split in two:
//   k_fast_score  : a streaming pass over each level.  One wavefront owns a (strip, segment) region of 248 x 24 pixels plus a
//                   one-pixel halo ring and walks down the rows: one aligned dword load per lane per row, kept at seven bits per pixel
//                   in a register ring of the last 7 rows (the screen) and as it is in a wavefront-private LDS ring of 14 + 7 mirrored
//                   rows whose every address is a per-block pointer + a compile-time offset (the exact test) -- no workgroup barriers.
//                   Every pixel is screened with four opposite ring pairs (any 9-arc contains one pixel of every opposite pair, of one
//                   polarity), four pixels per instruction in 32-bit arithmetic at seven bits per pixel (Screen4); the neighbour lanes'
//                   dwords come over the LDS crossbar (ds_bpermute).  The 7 - 12 % that pass are
//                   compacted into a wavefront-private LDS queue, drained oldest first, and the exact test runs on 64-lane batches:
//                   max over the 16 arcs of the arc minimum (branch free) is both the corner test (> t_min = min(fastTh, 7))
//                   and cornerScore + 1.  Corners (3-4 % of the pixels) are appended to a list in the wavefront's LDS -- no
//                   atomics, the count lives in a scalar register (a region with more corners than the list holds flushes it to
//                   its slice of an HBM list array).  At the end of the segment the wavefront does the in-cell 3x3 NMS itself
//                   on a byte tile laid over its LDS (neighbours outside the corner's own cell interior count as 0), compacts
//                   the survivors in place, marks cells that own a survivor >= fastTh and writes the survivors out: those that
//                   reach fastTh straight into the level's candidate array, the others into the level's low list (one
//                   atomic per class and wavefront reserves the slots).  HBM traffic: the level once in, the survivors out.
//   the per-cell vote (survivors >= fastTh if the cell has any, else the literal-7 fallback) needs every region of a cell to
//   be finished, so it is taken by the quad-tree kernel, which appends the low survivors of cells without a high one to the
//   level's candidates (octree.hip).  Candidate order in HBM is arbitrary: the quad-tree orders by coordinates.
//
// That is the SINGLE-PASS form (the level streams at min(fastTh, 7)).  When fastTh > 7 a level can instead take the THRESHOLD-
// ADAPTIVE TWO-PASS form: k_fast_score streams it at fastTh -- a corner >= fastTh can only be suppressed by a neighbour >= fastTh,
// so the survivors are exactly the first call's keypoints, and far fewer pixels reach the exact test -- and
//   k_fast_cells_list / k_fast_cells : redo the cells left without a keypoint at the literal 7, per cell, literally (the second call
//                   of :797), appending their survivors at the same cursor.
// tpass[level] (device memory, per pipeline lane) says which form a level takes in a batch; k_octree counts the batch's fall-back
// cells and k_assemble re-decides the levels for the lane's next batch (describe.hip: adapt_fast_mode).  Both forms give the same
// candidates; the parity tests force each (UVO_TUNE_FAST_MODE).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "common.hpp"
#include "fast_geom.hpp"

namespace uvo {

// the lane mask of a predicate, straight from the compare (HIP's __ballot(int) goes through a 0 / 1 integer and a second compare)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

#ifndef UVO_FC_DIRECT_MAX
#define UVO_FC_DIRECT_MAX 2  // batches up to this many frames redo their fall-back cells without a list (measured against always: see profiles/r04_fast_probes.txt)
#endif
#ifndef UVO_FAST_WAVES
#define UVO_FAST_WAVES 4       // wavefronts (regions) per workgroup
#endif
#ifndef UVO_FAST_MIN_BLOCKS
#define UVO_FAST_MIN_BLOCKS 5  // workgroups per CU the register allocation is held to; LDS: 31 KB per workgroup -> five fit (at
                               // exactly 32 KB only four do: measured, 0.87 instead of 0.76 ms per 256-frame launch)
#endif
#ifndef UVO_FAST_LIST_CAP
#define UVO_FAST_LIST_CAP 320
#endif
constexpr int FL_CAP = UVO_FAST_LIST_CAP;       // corner records a wavefront keeps in LDS (a 248 x 24 region of these frames holds ~250); a busier region
                                  // flushes its list to the region's slice of the HBM list array and carries on

// max over the 16 arcs of 9 contiguous ring pixels of min(d).  A 9-window always straddles the two 8-pixel halves of the
// ring, so with running minima towards the end of each half (S) and from the start of each half (P) every arc minimum is
// one more min: arc(k) = min(S[k], P[(k + 8) & 15]).  28 + 16 two-operand minima, then a max fold -- branch free.
// The differences fit 9 bits, so everything runs on 16-bit two-operand min / max: on gfx950 v_min_i16 / v_max_i16 issue in
// 2.4 cycles per wavefront, the 32-bit and three-operand forms in 4.4 (tools/ubench/valu_rate3.hip).
typedef short d16;
__device__ __forceinline__ d16 mn16(d16 a, d16 b) { return a < b ? a : b; }
__device__ __forceinline__ d16 mx16(d16 a, d16 b) { return a > b ? a : b; }
__device__ __forceinline__ d16 arc9_maxmin(const d16* d) {
  d16 S[16], P[16];
  S[7] = d[7], S[15] = d[15], P[0] = d[0], P[8] = d[8];
#pragma unroll
  for (int k = 6; k >= 0; --k) S[k] = mn16(d[k], S[k + 1]), S[k + 8] = mn16(d[k + 8], S[k + 9]);
#pragma unroll
  for (int k = 1; k < 8; ++k) P[k] = mn16(d[k], P[k - 1]), P[k + 8] = mn16(d[k + 8], P[k + 7]);
  d16 m9[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m9[k] = mn16(S[k], P[(k + 8) & 15]);
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; ++k) m9[k] = mx16(m9[k], m9[k + w]);
  return m9[0];
}

// min over the 16 arcs of max(d): the darker polarity without negating d (max over arcs of min(-d) = -this)
__device__ __forceinline__ d16 arc9_minmax(const d16* d) {
  d16 S[16], P[16];
  S[7] = d[7], S[15] = d[15], P[0] = d[0], P[8] = d[8];
#pragma unroll
  for (int k = 6; k >= 0; --k) S[k] = mx16(d[k], S[k + 1]), S[k + 8] = mx16(d[k + 8], S[k + 9]);
#pragma unroll
  for (int k = 1; k < 8; ++k) P[k] = mx16(d[k], P[k - 1]), P[k + 8] = mx16(d[k + 8], P[k + 7]);
  d16 m9[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m9[k] = mx16(S[k], P[(k + 8) & 15]);
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; ++k) m9[k] = mn16(m9[k], m9[k + w]);
  return m9[0];
}

// The screen: any 9-arc holds one pixel of every opposite ring pair, of one polarity -- so a corner has, in each of four opposite pairs,
// a member brighter than v + t, or in each a member darker than v - t.  Evaluated on all four pixels of a dword at once in plain 32-bit
// add / sub / and / or (the instructions gfx950 issues at twice the rate of packed 16-bit min / max), at seven bits per pixel: with x7 = x >> 1 and th = (t + 1) >> 1,
//   p > v + t  =>  p7 >= v7 + th      and      p < v - t  =>  p7 <= v7 - th
// (floor((a + b) / 2) >= floor(a / 2) + floor(b / 2)), so a test on the 7-bit values passes every pixel the exact test can pass -- and
// hardly any more (7.23 % instead of 7.19 % of the pixels of the benchmark frames at t = 20, 12.1 % instead of 10.7 % at t = 7).  Seven-bit
// values leave bit 7 of every byte free: a comparison is one addition or subtraction whose carry stops there.
//   bright:  bit 7 of p7 + (128 - min(v7 + th, 128))     <=>  p7 >= v7 + th     (sum <= 255: no carry into the next byte)
//   dark  :  bit 7 of max(128 + v7 - th, 127) - p7       <=>  p7 <= v7 - th     (minuend >= 127 >= p7: no borrow)
struct Screen4 {
  uint32_t cb, cd;  // per centre dword: the bright addend, the dark minuend
};
constexpr uint32_t kM7 = 0x7f7f7f7fu, kH7 = 0x80808080u;
__device__ __forceinline__ uint32_t seven(uint32_t x, uint32_t m7) { return (x >> 1) & m7; }
__device__ __forceinline__ Screen4 screen4_centre7(uint32_t v7, uint32_t thv, uint32_t m7, uint32_t h7) {
  const uint32_t w = v7 + thv;                         // <= 127 + 128: stays in its byte
  const uint32_t wh = w & h7, wl = wh - (wh >> 7);     // 0x7f in the bytes whose bit 7 is set
  const uint32_t u = (v7 | h7) - thv;                  // 128 + v7 - th >= 0
  const uint32_t uh = u & h7, ul = uh - (uh >> 7);
  Screen4 r;
  r.cb = h7 - (w & ~wl);                               // 128 - min(w, 128)
  r.cd = u | (m7 & ~ul);                               // max(u, 127)
  return r;
}

// LDS row ring of a wavefront.  The streaming loop is unrolled by seven rows (the register ring), so the ring's period is two blocks:
// a block of parity pb writes its row u to slot 7 pb + u, and the rows of the even blocks are mirrored into slots 14 .. 20.  The seven
// rows j-6 .. j around a centre are then always seven consecutive slots -- 8 + u .. 14 + u in an even block, 1 + u .. 7 + u in an odd one
// -- so that every LDS address of the loop is a per-block base + a compile-time offset: no scalar arithmetic per row.
constexpr int FR_SLOTS = 21;
constexpr int FR_PITCH = 66;        // row pitch in dwords: 64 + 2 so that the same column of consecutive rows hits different banks
constexpr int FW_RING_DW = FR_SLOTS * FR_PITCH;  // row ring, then the queue: one LDS block per wavefront
constexpr int FQ_CAP = (FS_ROWS_MAX + 2) * 64 - FW_RING_DW > 278 ? (FS_ROWS_MAX + 2) * 64 - FW_RING_DW : 278;  // the queue takes what the NMS tile leaves beside the ring (at least what it needs)
static_assert(FQ_CAP >= 64 + 128 + 16, "queue: < 64 left over + <= 128 pushed per half row (drained in between)");
constexpr int FW_DWORDS = FW_RING_DW + FQ_CAP;
constexpr int FT_PITCH = 256, FT_ROWS = FS_ROWS_MAX + 2;      // NMS score tile, laid over ring + queue at the end of the segment
static_assert(FT_PITCH * FT_ROWS <= FW_DWORDS * 4, "NMS tile must fit the wavefront's LDS block");

// out of line on purpose: the streaming loop below inlines the scoring chunk a dozen times, and this runs once in a blue moon
__device__ __noinline__ void flush_corner_list(const uint32_t* list, uint32_t* dst, int n, int lane) {
  for (int i = lane; i < n; i += 64) dst[i] = list[i];
}

// max(brighter, darker) arc strength of the pixel whose (-3 rows, -3 columns) corner is rm3 in a byte tile of row pitch RB:
// the pixel is a FAST-9 corner at threshold t iff the result is > t, and cornerScore = result - 1.
template <int RB>
__device__ __forceinline__ int ring_strength(const uint8_t* rm3) {
  const uint8_t *rm2 = rm3 + RB, *rm1 = rm3 + 2 * RB, *r0 = rm3 + 3 * RB, *rp1 = rm3 + 4 * RB, *rp2 = rm3 + 5 * RB, *rp3 = rm3 + 6 * RB;
  const d16 v = (d16)r0[3];
  d16 d[16];
  d[0] = rp3[3], d[1] = rp3[4], d[2] = rp2[5], d[3] = rp1[6], d[4] = r0[6], d[5] = rm1[6], d[6] = rm2[5], d[7] = rm3[4];
  d[8] = rm3[3], d[9] = rm3[2], d[10] = rm2[1], d[11] = rm1[0], d[12] = r0[0], d[13] = rp1[0], d[14] = rp2[1], d[15] = rp3[2];
  // corner at t  <=>  some 9-arc has all diffs > t (brighter) or all < -t (darker)  <=>  max(sb, sd) > t;
  // cornerScore = max(sb, sd) - 1.  No masks, no divergent branches.  The centre is the same for all sixteen differences, so the arc
  // minima / maxima are taken over the ring pixels themselves and the centre comes off once: sb = max over arcs of min(p) - v,
  // sd = v - min over arcs of max(p) -- two subtractions instead of sixteen.
  return max((int)arc9_maxmin(d) - (int)v, (int)v - (int)arc9_minmax(d));
}

// Full segment test + cornerScore of the queued pixels [first, first+count), one per lane.  The 16 ring pixels are read
// back from the wavefront's LDS row ring: the queue entry carries the byte address of the pixel in the ring (the seven rows around
// any centre are consecutive slots, see FR_SLOTS, and every read is base + immediate).
// entry = byte address of (pixel - 3 rows - 3 columns) in the workgroup's LDS block | xl << 15 | (row - py0 + 1) << 23.  Corners go to the wavefront's list as xl | row' << 8 | score << 16.
__device__ __forceinline__ void fast_score_chunk(const uint8_t* rows, uint32_t first_byte, int count, int lane, int t_min,
                                                 uint32_t* list, uint32_t* __restrict__ region, int& ncorner, int& nflushed) {
  bool corner = false;
  uint32_t packed = 0;
  if (lane < count) {
    const uint32_t meta = *reinterpret_cast<const uint32_t*>(rows + first_byte + lane * 4);  // the queue lives in the same LDS block
    const int xl = (int)((meta >> 15) & 0xff), rrp = (int)(meta >> 23);
    // the entry's address field points 3 rows above and 3 bytes left of the pixel, relative to the workgroup's LDS block: every ring
    // pixel is that one register + an immediate offset
    const int best = ring_strength<FR_PITCH * 4>(rows + (meta & 0x7fffu));
    // a score of 0 (only possible at t_min = 0) can never survive NMS nor suppress anything: one compare against max(t_min, 1)
    corner = best > (t_min > 1 ? t_min : 1);
    packed = (uint32_t)xl | ((uint32_t)rrp << 8) | ((uint32_t)(best - 1) << 16);
  }
  const uint64_t m = ballot64(corner);
  if (m) {
    const int add = (int)__popcll(m);
    if (ncorner + add > FL_CAP) {  // wave-uniform, rare: the LDS list is flushed to the region's slice in memory and starts again
      flush_corner_list(list, region + nflushed, ncorner, lane);
      nflushed += ncorner;
      ncorner = 0;
    }
    if (corner) list[ncorner + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = packed;
    ncorner += add;
  }
}

// One wavefront per (strip, segment) region: 248 x rows_per_seg pixels of the detection window plus a one-pixel halo ring whose
// scores are computed redundantly (the ring radius is 3 and the strip's halo lanes are 4 px wide, so no extra loads), so that the
// in-cell 3x3 non-max suppression can be done by the wavefront itself at the end of the segment: the corner scores are scattered
// into a dense byte tile that reuses the LDS of the row ring + queue, the eight neighbours of every corner are read from it
// (neighbours outside the corner's own FAST cell count as 0), survivors are compacted in place and mark their cell when they
// reach fastTh.  No score plane in HBM, no zero fill, no second gather pass.
__global__ __launch_bounds__(64 * UVO_FAST_WAVES, UVO_FAST_MIN_BLOCKS) void k_fast_score(const uint8_t* __restrict__ pyr, int64_t pyr_block, FastLevels L, const int32_t* __restrict__ tpass, int fast_th,
                                                    uint32_t* __restrict__ cor, uint8_t* __restrict__ cell_hi, uint32_t* __restrict__ cand_xy,
                                                    uint32_t* __restrict__ cand_sc, uint32_t* __restrict__ cand_lo, int64_t cand_block,
                                                    int32_t* __restrict__ cursor, Level0View l0) {
  __shared__ uint32_t s_mem[UVO_FAST_WAVES][FW_DWORDS];
  __shared__ uint32_t s_list[UVO_FAST_WAVES][FL_CAP];
  const int wv = wave_in_block(), lane = threadIdx.x & 63;
  uint32_t* rows32 = s_mem[wv];
  // work item (one per wavefront) -> (level, strip, segment); window = padded cols [32, w) x rows [32, h)
  const int vb = xcd_contiguous((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int item = (vb % (int)gridDim.x) * UVO_FAST_WAVES + wv, f = vb / (int)gridDim.x;  // an XCD walks whole frames, region after region
  int level, X0, py0, nsub;
  if (!fast_region(L, item, level, X0, py0, nsub)) return;
  const FastLevel g = L.l[level];
  // threshold of this level's streaming pass (wave-uniform): fastTh when the level runs threshold-adaptive -- the cells left without a
  // survivor are then redone at the literal 7 by k_fast_cells -- or min(fastTh, 7), one pass that keeps the low survivors for the vote
  const int t_min = tpass[level];
  const int64_t region_id = (int64_t)f * L.items_per_frame + item;
  uint32_t* region = cor + region_id * (int64_t)FS_REGION_ENTRIES;  // only touched when the LDS list overflows
  uint32_t* list = s_list[wv];
  int ncorner = 0;   // corners in the LDS list
  int nflushed = 0;  // corners already moved to the region's slice in memory
  // level 0 may be the caller's image read in place: the detection window and its 3-pixel ring lie inside the image, so only the base,
  // the pitch and the two clamps change (wave-uniform)
  const bool ip = level == 0 && l0.vbase != nullptr;
  const uint8_t* src = ip ? l0.vbase + f * l0.frame_stride : pyr + f * pyr_block + g.plane_off;
  const int pitch = ip ? l0.pitch : g.pitch;
  // nsub sub-strips side by side (1: the whole wavefront; 2 / 4: 32 / 16 lanes each, consecutive row segments of one narrow strip).
  // Everything that depends on the row is kept relative to the sub-strip's own first row, so the loop below stays uniform.
  const int lps = 64 / nsub;                 // lanes per sub-strip
  const int sub = (lane * nsub) >> 6, ls = lane - sub * lps;
  const int sub_px = 4 * lps;                // pixels a sub-strip spans, halo lanes included
  const int sub_shift = nsub == 1 ? 8 : (nsub == 2 ? 7 : 6);  // sub_px = 1 << sub_shift
  const int py0l = py0 + sub * L.rows_per_seg;
  const int nrows_l = max(min(py0l + L.rows_per_seg, g.h) - py0l, 0);  // 0: this sub-strip lies below the level
  const int X = X0 + ls * 4;  // padded column of the lane's first pixel; the first and last lane of a sub-strip are its halo
  const int x_last = ip ? g.w + kPad - 4 : g.pitch - 4;  // last dword of a row that may be loaded (in place: the image's last four columns)
  const int Xc = X > x_last ? x_last : X;
  const int nrows = min(py0 + L.rows_per_seg, g.h) - py0;  // sub-strip 0 has the most rows
  const int nsrc = nrows + 8;  // centre rows py0-1 .. py1 need source rows py0-4 .. py1+3
  // pixel K of the lane is screened when it lies in the sub-strip or its one-pixel halo and in the detection window
  bool okv[4];
#pragma unroll
  for (int K = 0; K < 4; ++K) {
    const int xs = ls * 4 + K;
    okv[K] = nrows_l > 0 && xs >= 3 && xs <= sub_px - 4 && X + K >= 32 && X + K < g.w;
  }
  // bit 7 of byte K: the lane screens its pixel K
  uint32_t mkall = (okv[0] ? 0x00000080u : 0u) | (okv[1] ? 0x00008000u : 0u) | (okv[2] ? 0x00800000u : 0u) | (okv[3] ? 0x80000000u : 0u);
  // constants of the four-pixel screen, in vector registers (a fast-class instruction that reads a scalar register or a literal is not one)
  uint32_t m7 = kM7, h7 = kH7, thv = (uint32_t)((t_min + 1) >> 1) * 0x01010101u;
  asm volatile("" : "+v"(m7), "+v"(h7), "+v"(thv));
  // bit rrp: the lane screens centre row rrp of its sub-strip (0 = halo row above, nrows_l + 1 = halo row below) -- the row lies in the
  // detection window, pcl = py0l - 1 + rrp in [32, g.h).  One bit-field extract per row instead of three compares and two selects.
  uint32_t rowbits;
  {
    const int lo = max(33 - py0l, 0), hi = min(min(nrows_l + 1, g.h - py0l), 31);
    rowbits = hi >= lo ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
  }

  // Row loads run one unrolled block (7 rows) ahead of their use so that a wavefront never waits on the row it needs next.
  const int last_row = g.h + 15;  // last row of the padded plane
  auto load_row = [&](int j) -> uint32_t {
    const int r = min(py0l - 4 + j, last_row);  // rows past the plane belong to a sub-strip below the level: never used
    // (a plane is far below 4 GB: wave-uniform base + a 32-bit lane offset by a 24-bit multiply -- the 64-bit multiply-add the pointer
    // arithmetic asks for issues at a quarter of the rate)
    return *reinterpret_cast<const uint32_t*>(src + (__umul24((uint32_t)r, (uint32_t)pitch) + (uint32_t)Xc));
  };
  // queue entry of the lane's pixel K in row u of a block = ent_blk + u * kEntRow + K * kEntPix:
  //   address part: this wavefront's ring + the window's first slot (row j - 6) + the lane's pixel K - 3 columns; xl part; row part
  constexpr uint32_t kEntRow = (uint32_t)(FR_PITCH * 4) + (1u << 23), kEntPix = (1u << 15) + 1u;
  const uint32_t lane_entry = (uint32_t)(wv * FW_DWORDS * 4 + lane * 4 - 3) + ((uint32_t)(lane * 4) << 15);
  const uint32_t q0 = (uint32_t)((wv * FW_DWORDS + FW_RING_DW) * 4);  // byte address of the queue in the workgroup's LDS block
  uint32_t qa = q0;                                                    // ... of its first free entry (wavefront-uniform)
  uint8_t* lds8 = reinterpret_cast<uint8_t*>(&s_mem[0][0]);
  uint32_t qold = 0;  // bytes at the bottom of the queue that were already there at the last checkpoint
  // the 64 oldest entries are scored, the (< 128) younger ones move down to the bottom of the queue
  auto drain_oldest = [&]() {
    fast_score_chunk(lds8, q0, 64, lane, t_min, list, region, ncorner, nflushed);
    uint32_t* qq = reinterpret_cast<uint32_t*>(lds8 + q0);
    const uint32_t rem = qa - q0 - 256u, l4 = (uint32_t)lane * 4u;
    const uint32_t e0 = l4 < rem ? qq[64 + lane] : 0u, e1 = l4 + 256u < rem ? qq[128 + lane] : 0u;
    if (l4 < rem) qq[lane] = e0;
    if (l4 + 256u < rem) qq[64 + lane] = e1;
    qa -= 256u;
    qold = qold > 256u ? qold - 256u : 0u;
  };
  uint32_t nxt[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) nxt[u] = load_row(u);  // unconditional (the row index is clamped into the plane): with a branch around a load
                                                     // the compiler cannot count the loads in flight and waits for all of them
  // Register rings of the last seven rows, at seven bits per pixel (the screen never looks at bit 0): the row itself and its views two
  // pixels to the right / left (ring pixels 2, 6 / 14, 10 of the centres two rows above and below).  The neighbour lanes' dwords come
  // over the LDS crossbar (ds_bpermute: no LDS memory, no vector-ALU slot), from the lanes next to it in the wavefront whatever sub-strip
  // they belong to (63 and 0 are neighbours): the first lane of a sub-strip screens its pixel 3 and the last its pixel 0, and the bytes
  // of the views those two read are the lane's own -- V2p of the lane before holds them because ITS right neighbour is this lane.
  uint32_t S7[7], V2p[7];
  const int lm4 = ((lane + 63) & 63) * 4, lp4 = ((lane + 1) & 63) * 4;
  uint32_t kpix = kEntPix, kentrow = kEntRow;
  uint32_t qdump = q0 + (uint32_t)(FQ_CAP - 1) * 4u;  // a dword of the queue no entry ever reaches (< 64 + 128 entries at any time)
  asm volatile("" : "+v"(qdump));
  asm volatile("" : "+v"(mkall), "+v"(kpix), "+v"(kentrow));
  int pbo = 0;  // 7 * parity of the block
  for (int base = 0; base < nsrc; base += 7, pbo ^= 7) {
    uint32_t cur[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) cur[u] = nxt[u];
#pragma unroll
    for (int u = 0; u < 7; ++u) nxt[u] = load_row(base + 7 + u);
    // everything of the block that depends on its parity or its first row, once: the rows below only add compile-time offsets
    uint32_t* wr = rows32 + pbo * FR_PITCH + lane;          // slot 7 pb + u
    uint32_t* wr2 = rows32 + (14 - pbo) * FR_PITCH + lane;  // its mirror 14 + u in an even block (an odd block stores the same slot twice)
    const uint32_t rowbits_blk = base ? rowbits >> (base - 6) : rowbits << 6;  // bit u: the lane screens the centre row of loop row base + u
    const uint32_t ent_blk = lane_entry + ((uint32_t)((8 - pbo) * FR_PITCH * 4) + ((uint32_t)(base - 6) << 23));
    uint32_t ent_row = ent_blk;  // entry of the lane's pixel 0 in the row at hand
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int j = base + u;
      if (j < nsrc) {
        const uint32_t C = cur[u];
        wr[u * FR_PITCH] = C;
        wr2[u * FR_PITCH] = C;
        const uint32_t c7 = seven(C, m7);
        S7[u] = c7;
        V2p[u] = __builtin_amdgcn_alignbyte((uint32_t)__builtin_amdgcn_ds_bpermute(lp4, (int)c7), c7, 2);
        if (j >= 6) {
          // centre row jc = j - 3, relative to the sub-strip rrp = j - 6 (0 = halo row above, nrows + 1 = halo row below); rows jc-3 .. jc+3
          // sit in register slots (u+1)%7 .. u and in the ring slots the window pointers start at
          const uint32_t rowm = (uint32_t)__builtin_amdgcn_sbfe((int)rowbits_blk, (uint32_t)u, 1u);  // 0 / ~0: this lane screens this row
          {
            const int sm3 = (u + 1) % 7, sm2 = (u + 2) % 7, s0 = (u + 4) % 7, sp2 = (u + 6) % 7, sp3 = u;
            const uint32_t cc = S7[s0];
            const uint32_t Lc = (uint32_t)__builtin_amdgcn_ds_bpermute(lm4, (int)cc), Rc = (uint32_t)__builtin_amdgcn_ds_bpermute(lp4, (int)cc);
            // a lane's view two pixels to the left is its left neighbour's view two pixels to the right
            const uint32_t q2 = V2p[sp2], q6 = V2p[sm2];
            const uint32_t q14 = (uint32_t)__builtin_amdgcn_ds_bpermute(lm4, (int)q2), q10 = (uint32_t)__builtin_amdgcn_ds_bpermute(lm4, (int)q6);
            const Screen4 sc = screen4_centre7(cc, thv, m7, h7);
            const uint32_t q0_ = S7[sp3], q8 = S7[sm3], q4 = __builtin_amdgcn_alignbyte(Rc, cc, 3), q12 = __builtin_amdgcn_alignbyte(cc, Lc, 1);
            const uint32_t bright = ((q0_ + sc.cb) | (q8 + sc.cb)) & ((q4 + sc.cb) | (q12 + sc.cb)) & ((q2 + sc.cb) | (q10 + sc.cb)) & ((q6 + sc.cb) | (q14 + sc.cb));
            const uint32_t dark = ((sc.cd - q0_) | (sc.cd - q8)) & ((sc.cd - q4) | (sc.cd - q12)) & ((sc.cd - q2) | (sc.cd - q10)) & ((sc.cd - q6) | (sc.cd - q14));
            const uint32_t pm = (bright | dark) & rowm & mkall;  // bit 7 of byte K: the lane's pixel K passes
            // The flag of pixel K is the sign of byte K: one compare with a sign-extending byte select gives the lane mask, which is used as
            // it is for the store's EXEC and for the ranks (left to the compiler, `(pm & bit) != 0` becomes two compares and an AND).
            const uint32_t e1 = ent_row + kpix, e2 = e1 + kpix, e3 = e2 + kpix;
#define UVO_FAST_PUSHB(BYTE, ENTRY)                                                                                                     \
  {                                                                                                                                     \
    uint64_t m;                                                                                                                         \
    asm("v_cmp_gt_i32_sdwa %0, 0, sext(%1) src0_sel:DWORD src1_sel:BYTE_" #BYTE : "=s"(m) : "v"(pm));                                   \
    /* no EXEC region: the lanes that do not pass store to a dword of the queue no entry ever reaches (a select costs the SIMD less    \
       than the two scalar instructions that open and close a region: DESIGN.md section 7.2) */                                         \
    const uint32_t a_ = qa + 4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));           \
    *reinterpret_cast<uint32_t*>(lds8 + (__builtin_amdgcn_inverse_ballot_w64(m) ? a_ : qdump)) = (ENTRY);                               \
    const uint32_t n_ = (uint32_t)__popcll(m);                                                                                          \
    asm volatile("s_lshl2_add_u32 %0, %1, %0" : "+s"(qa) : "s"(n_) : "scc"); /* the end moves now, on the scalar unit, in one op */      \
  }
            UVO_FAST_PUSHB(0, ent_row)
            UVO_FAST_PUSHB(1, e1)
            while (qa >= q0 + 256u) drain_oldest();  // keeps the queue within FQ_CAP
            UVO_FAST_PUSHB(2, e2)
            UVO_FAST_PUSHB(3, e3)
#undef UVO_FAST_PUSHB
          }
          // ---- drain full batches, oldest first.  What stays behind is younger than what left, and only what has stayed behind for two
          // checkpoints in a row (after rows 2 and 6 of a block: 7 rows apart) is scored as an incomplete batch.  A queued pixel's oldest
          // ring row (j - 6) is overwritten 14 rows after it was written, i.e. 8 rows after the pixel was queued; it waits <= 6.
          // (Three generations of checkpoints two rows apart -- an incomplete batch only after 4 - 6 quiet rows -- issue the same number of
          // instructions: on these frames the incomplete batches that are left belong to sparse strips and pyramid levels.) ----
          while (qa >= q0 + 256u) drain_oldest();
          if (u == 2 || u == 6) {
            if (qold) {
              fast_score_chunk(lds8, q0, (int)((qa - q0) >> 2), lane, t_min, list, region, ncorner, nflushed);
              qa = q0;
            }
            qold = qa - q0;
          }
        }
        ent_row += kentrow;
      }
    }
  }
  if (qa > q0) fast_score_chunk(lds8, q0, (int)((qa - q0) >> 2), lane, t_min, list, region, ncorner, nflushed);

  // ---- in-cell 3x3 non-max suppression of the region's corners (cv::FAST with nonmaxSuppression on the cell ROI) ----
  uint8_t* tile = reinterpret_cast<uint8_t*>(rows32);  // [row' = row - py0 + 1][xl], FT_PITCH bytes per row; ring and queue are dead
#pragma unroll
  for (int i = 0; i < FT_ROWS * FT_PITCH / 4 / 64; ++i) rows32[i * 64 + lane] = 0u;  // 26 stores, no loop bookkeeping on the scalar unit
  uint8_t* hi = cell_hi + (int64_t)f * L.flags_per_frame + g.flag_base;
  // the corner list (in LDS, or in memory once it has spilled) is compacted in place to the NMS survivors: x | y << 12 | score << 24
  // relative to (minBorder, minBorder)
  auto nms_list = [&](uint32_t* C) -> int {
    __builtin_amdgcn_wave_barrier();
    constexpr int NB = 4;  // batches of 64 corners handled together: their LDS round trips overlap
    for (int base = 0; base < ncorner; base += 64 * NB) {
      uint32_t e[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) e[u] = base + 64 * u + lane < ncorner ? C[base + 64 * u + lane] : 0u;
#pragma unroll
      for (int u = 0; u < NB; ++u)
        if (base + 64 * u + lane < ncorner) tile[((e[u] >> 8) & 0xff) * FT_PITCH + (e[u] & 0xff)] = (uint8_t)(e[u] >> 16);
    }
    __builtin_amdgcn_wave_barrier();
    int nkeep = 0;
    for (int base = 0; base < ncorner; base += 64 * NB) {
      uint32_t e[NB], out[NB];
      bool keep[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) e[u] = base + 64 * u + lane < ncorner ? C[base + 64 * u + lane] : 0u;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const bool valid = base + 64 * u + lane < ncorner;
        keep[u] = false, out[u] = 0;
        const int xl = (int)(e[u] & 0xff), rrp = (int)((e[u] >> 8) & 0xff), ss = (int)(e[u] >> 16);
        // the sub-strip the corner belongs to: columns [es * sub_px, (es + 1) * sub_px) of the tile, rows of segment es (sub_px = 256 / nsub:
        // shifts and 24-bit multiplies -- a 32-bit integer multiply issues at a quarter of their rate)
        const int es = xl >> sub_shift;
        const int xs = xl & (sub_px - 1), py0e = py0 + (int)__umul24((uint32_t)es, (uint32_t)L.rows_per_seg);
        const int nrows_e = min(py0e + L.rows_per_seg, g.h) - py0e;
        const bool owned = valid && xs >= 4 && xs < sub_px - 4 && rrp >= 1 && rrp <= nrows_e;  // not a halo-ring corner
        if (owned) {
          // coordinates relative to (minBorder, minBorder), as the candidate list wants them
          const int xr = X0 + xs - kPad - kMinBorder, yr = py0e + rrp - 1 - kPad - kMinBorder;
          // (xr-3) / wCell, (yr-3) / hCell by a 24-bit multiply with ceil(2^24 / cell): exact for dividends below 4096 and cells of 17 .. 66 pixels
          int cj = (int)(__umul24((uint32_t)(xr - 3), g.inv_wcell) >> 24), ci = (int)(__umul24((uint32_t)(yr - 3), g.inv_hcell) >> 24);
          cj = cj > g.nCols - 1 ? g.nCols - 1 : cj;
          ci = ci > g.nRows - 1 ? g.nRows - 1 : ci;
          // interior of the owning cell: [j*wCell + 3, min(j*wCell + wCell + 6, bw) - 3) and the same in y.  The corner lies inside it, so
          // a neighbour can only fall outside on the side where the corner touches the interior's edge
          const int cx0 = (int)__umul24((uint32_t)cj, (uint32_t)g.wCell) + 3, cx1 = min(cx0 + g.wCell + 3, g.bw) - 3;
          const int cy0 = (int)__umul24((uint32_t)ci, (uint32_t)g.hCell) + 3, cy1 = min(cy0 + g.hCell + 3, g.bh) - 3;
          // all eight neighbours are read (an owned corner's are inside the tile) and the ones beyond the cell's interior masked out: one
          // EXEC region, eight loads in flight, instead of a predicated load each
          const uint32_t mL = xr > cx0 ? 0xffu : 0u, mR = xr + 1 < cx1 ? 0xffu : 0u, mU = yr > cy0 ? 0xffu : 0u, mD = yr + 1 < cy1 ? 0xffu : 0u;
          const uint8_t* c = tile + rrp * FT_PITCH + xl;
          const uint32_t n0 = c[-FT_PITCH - 1], n1 = c[-FT_PITCH], n2 = c[-FT_PITCH + 1], n3 = c[-1], n4 = c[1], n5 = c[FT_PITCH - 1], n6 = c[FT_PITCH],
                         n7 = c[FT_PITCH + 1];
          const uint32_t up = max(max(n0 & mL, n1), n2 & mR) & mU, dn = max(max(n5 & mL, n6), n7 & mR) & mD, mid = max(n3 & mL, n4 & mR);
          const bool k = (uint32_t)ss > max(max(up, mid), dn);
          keep[u] = k;
          if (k && ss >= fast_th) hi[(int)__umul24((uint32_t)ci, (uint32_t)g.nCols) + cj] = 1;  // idempotent plain store: every writer stores the same value
          out[u] = (uint32_t)xr | ((uint32_t)yr << 12) | ((uint32_t)ss << 24);
        }
      }
      // in-place compaction: survivors land at or before the start of the group that was just read
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const uint64_t m = ballot64(keep[u]);
        if (m) {
          if (keep[u]) C[nkeep + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = out[u];
          nkeep += (int)__popcll(m);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    return nkeep;
  };
  // Survivors leave the wavefront in two classes (the per-cell threshold vote of src/ORBextractor.cc:792-799 needs every region of a
  // cell to be finished, so it is taken by the quad-tree kernel): those that reach fastTh are candidates whatever the vote says and go
  // straight to the level's candidate array; the rest (7 <= score < fastTh) wait in the level's "low" list for the vote.  One atomic
  // per class and wavefront reserves the slots; the order inside the arrays is arbitrary (the quad-tree orders by coordinates).
  auto emit = [&](uint32_t* C, int nkeep) {
    int n_hi = 0;
    for (int base = 0; base < nkeep; base += 64) {
      const bool valid = base + lane < nkeep;
      const uint32_t e = valid ? C[base + lane] : 0u;
      n_hi += (int)__popcll(ballot64(valid && (int)(e >> 24) >= fast_th));
    }
    const int n_lo = nkeep - n_hi;
    int32_t* cur = cursor + 2 * ((int64_t)f * L.nlevels + level);
    int off_hi = 0, off_lo = 0;
    if (lane == 0) {
      if (n_hi) off_hi = atomicAdd(&cur[0], n_hi);
      if (n_lo) off_lo = atomicAdd(&cur[1], n_lo);
    }
    off_hi = __builtin_amdgcn_readfirstlane(off_hi), off_lo = __builtin_amdgcn_readfirstlane(off_lo);
    const int64_t co = (int64_t)f * cand_block + g.cand_off;
    for (int base = 0; base < nkeep; base += 64) {
      const bool valid = base + lane < nkeep;
      const uint32_t e = valid ? C[base + lane] : 0u;
      const bool is_hi = valid && (int)(e >> 24) >= fast_th, is_lo = valid && !is_hi;
      const uint64_t mh = ballot64(is_hi), ml = ballot64(is_lo);
      if (is_hi) {
        const int pos = off_hi + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u));
        if (pos < g.cand_cap) {
          cand_xy[co + pos] = (e & 0xfffu) | (((e >> 12) & 0xfffu) << 16);
          cand_sc[co + pos] = e >> 24;
        }
      } else if (is_lo) {
        const int pos = off_lo + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u));
        if (pos < g.cand_cap) cand_lo[co + pos] = e;
      }
      off_hi += (int)__popcll(mh), off_lo += (int)__popcll(ml);
    }
  };
  // two instances of the tail, so that each knows its address space: the list still in LDS (a quiet region), or in the region's slice
  // of memory once it has overflowed (the rule on corner-rich frames) -- a generic pointer would make every access a flat one
  if (nflushed) {
    for (int i = lane; i < ncorner; i += 64) region[nflushed + i] = list[i];
    ncorner += nflushed;
    __threadfence_block();  // the list in memory was written by this wavefront's own global stores
    const int nkeep = nms_list(region);
    __threadfence_block();
    emit(region, nkeep);
  } else {
    const int nkeep = nms_list(list);
    emit(list, nkeep);
  }
}


// ---- the sparse second pass of the threshold-adaptive form: FAST(cellROI, 7, nms) for the cells the streaming pass left empty ----
// src/ORBextractor.cc:792-799: `FAST(cell, kps, fastTh, true); if (kps.empty()) FAST(cell, kps, 7, true);`.  When a level's streaming
// pass ran at fastTh (tpass[level] > 7) its NMS survivors are exactly the first call's keypoints (a corner >= fastTh can only be
// suppressed by a neighbour >= fastTh), and cell_hi marks the cells that own one.  Every other cell is redone here, literally: the
// scores of ALL interior pixels at threshold 7 -- also those >= fastTh that annihilated each other in the first call; they still
// suppress their weaker neighbours in the second -- then the 3x3 suppression inside the cell's interior.  One wavefront per cell,
// `cpw` cells looked at per wavefront (most are skipped: a flag read and a ballot); survivors are appended to the level's
// candidates at the same cursor the streaming pass used.
constexpr int FC_WAVES = 4;
constexpr int FC_QCAP = 128;   // < 64 left over + <= 64 pushed per step
// LDS geometry of a wavefront, two sizes: cells up to 48 x 48 (every level whose detection window is at least 90 px wide and high: 3
// or more cells across, each at most 40 + 6) leave room for seven workgroups per CU; cells up to 66 x 66 (build_geom refuses larger
// ones) for three.  The kernel is a chain of short dependent phases per cell: wavefronts in flight are what it runs on.
template <int MAXROI>
struct FcGeom {
  static constexpr int TPITCH = (MAXROI + 3 + 3) / 4 * 4;   // ROI row pitch in bytes: the ROI + up to 3 bytes of dword alignment, whole dwords
  static constexpr int ND = TPITCH / 4;
  static constexpr int SPITCH = (MAXROI - 6 + 2 + 3) / 4 * 4, SROWS = MAXROI - 6 + 2;   // score tile: the interior inside a ring of zeros
  static constexpr int TILE_DW = MAXROI * ND;               // also holds the survivor list (<= (MAXROI - 6)^2 / 4 words)
  static constexpr int CCAP = MAXROI <= 48 ? 256 : 512;     // corner positions remembered per cell (16 bit each); a busier cell scans its whole score tile
  static_assert(TILE_DW >= ((MAXROI - 5) / 2) * ((MAXROI - 5) / 2), "survivor list must fit the ROI tile");
  static_assert(SROWS <= 64 && SPITCH <= 64, "corner positions are packed as ix | iy << 6");
};

// k_fast_cells_list: one thread per entry of a frame's cell-flag array (coalesced byte reads; flag_cell maps the entry back to its cell
// and level, -1 where the grid position is no cell) -- is the level threshold-adaptive and did the streaming pass leave the cell empty?
// Those cells are appended to `list` as (cell, frame), one reservation per workgroup; n_list counts them (zeroed again by k_octree,
// which runs behind both kernels).
constexpr int FCL_THREADS = 1024;
__global__ __launch_bounds__(FCL_THREADS) void k_fast_cells_list(const int32_t* __restrict__ flag_cell, int flags_per_frame, int nlevels, const int32_t* __restrict__ tpass,
                                                               const uint8_t* __restrict__ cell_hi, uint2* __restrict__ list, int32_t* __restrict__ n_list,
                                                               int list_cap) {
  __shared__ int s_n, s_base;
  __shared__ uint32_t s_adaptive;  // bit l: level l is threshold-adaptive in this batch
  const int idx = (int)(blockIdx.x * FCL_THREADS + threadIdx.x), f = blockIdx.y, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    uint32_t m = 0;
    for (int l = 0; l < nlevels; ++l) m |= tpass[l] > 7 ? 1u << l : 0u;
    s_adaptive = m, s_n = 0;
  }
  int fc = -1;
  uint8_t flag = 1;
  if (idx < flags_per_frame) fc = flag_cell[idx], flag = cell_hi[(int64_t)f * flags_per_frame + idx];
  __syncthreads();
  const bool todo = fc >= 0 && flag == 0 && ((s_adaptive >> (fc >> 24)) & 1u);
  const uint64_t m = ballot64(todo);
  int at = 0;
  if (m) {
    if (lane == 0) at = atomicAdd(&s_n, (int)__popcll(m));
    at = __builtin_amdgcn_readfirstlane(at) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  }
  __syncthreads();
  if (s_n == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(n_list, s_n);
  __syncthreads();
  if (todo && s_base + at < list_cap) list[s_base + at] = make_uint2((uint32_t)(fc & 0xffffff), (uint32_t)f);  // (the list holds every cell of the batch: the bound only bites if the counter was left stale)
}

// k_fast_cells: a fixed grid of wavefronts shares the listed cells evenly (fall-back cells cluster in the smooth parts of a frame: dealt
// by position, some wavefronts would redo a dozen cells in a row while most find none).
// With DIRECT (one or two frames: a latency call, nothing to balance) there is no list: wavefront w of frame blockIdx.y looks at entry w of
// the frame's cell-flag array itself and redoes the cell if it has to -- one launch instead of two.
template <int MAXROI, bool DIRECT>
__global__ __launch_bounds__(64 * FC_WAVES) void k_fast_cells(const uint8_t* __restrict__ pyr, int64_t pyr_block, FastLevels L, const CellDesc* __restrict__ cells,
                                                            const uint2* __restrict__ list, const int32_t* __restrict__ n_list,
                                                            const int32_t* __restrict__ flag_cell, const int32_t* __restrict__ tpass,
                                                            const uint8_t* __restrict__ cell_hi,
                                                            uint32_t* __restrict__ cand_xy, uint32_t* __restrict__ cand_sc, int64_t cand_block,
                                                            int32_t* __restrict__ cursor, Level0View l0) {
  typedef FcGeom<MAXROI> G;
  constexpr int FC_TPITCH = G::TPITCH, FC_ND = G::ND, FC_SPITCH = G::SPITCH, FC_SROWS = G::SROWS, FC_CCAP = G::CCAP;
  __shared__ uint32_t s_tile[FC_WAVES][G::TILE_DW];
  __shared__ uint32_t s_score[FC_WAVES][FC_SROWS * FC_SPITCH / 4];
  __shared__ uint32_t s_q[FC_WAVES][FC_QCAP];
  __shared__ uint16_t s_corner[FC_WAVES][FC_CCAP];
  const int wv = wave_in_block(), lane = threadIdx.x & 63;
  const int wave_id = (int)blockIdx.x * FC_WAVES + wv, n_waves = DIRECT ? 1 : (int)gridDim.x * FC_WAVES;
  int n_items, direct_cell = 0;
  if (DIRECT) {
    if (wave_id >= L.flags_per_frame) return;
    const int fc = flag_cell[wave_id];
    if (fc < 0 || tpass[fc >> 24] <= 7 || cell_hi[(int64_t)blockIdx.y * L.flags_per_frame + wave_id] != 0) return;
    direct_cell = fc & 0xffffff;
    n_items = wave_id + 1;  // exactly one trip through the loop below
  } else {
    n_items = *n_list;
    if (wave_id >= n_items) return;
  }
  uint32_t* tile32 = s_tile[wv];
  const uint8_t* tile8 = reinterpret_cast<const uint8_t*>(tile32);
  uint32_t* score32 = s_score[wv];
  uint8_t* score8 = reinterpret_cast<uint8_t*>(score32);
  uint32_t* q = s_q[wv];
  uint16_t* clist = s_corner[wv];
  // the score tile is zero between cells: cleared once here, afterwards every cell wipes the corners it wrote
  for (int i = lane; i < FC_SROWS * FC_SPITCH / 4; i += 64) score32[i] = 0u;
  for (int item = wave_id; item < n_items; item += n_waves) {
    const uint2 it = DIRECT ? make_uint2((uint32_t)direct_cell, blockIdx.y) : list[item];
    const int cell = __builtin_amdgcn_readfirstlane((int)it.x), f = __builtin_amdgcn_readfirstlane((int)it.y);
    const CellDesc c = cells[cell];
    const int level = __builtin_amdgcn_readfirstlane((int)c.level);
    const int rw = __builtin_amdgcn_readfirstlane((int)c.rw), rh = __builtin_amdgcn_readfirstlane((int)c.rh);
    const int x0 = __builtin_amdgcn_readfirstlane((int)c.x0), y0 = __builtin_amdgcn_readfirstlane((int)c.y0);
    const int ox = __builtin_amdgcn_readfirstlane((int)c.ox), oy = __builtin_amdgcn_readfirstlane((int)c.oy);
    const FastLevel g = L.l[level];
    const int iw = rw - 6, ih = rh - 6, npix = iw * ih;
    (void)ih;
    // ---- the ROI into LDS as aligned dwords: padded plane columns [a0, a0 + 4 nd), `sh` bytes of slack in front ----
    const int px0 = kPad + x0, py0 = kPad + y0, a0 = px0 & ~3, sh = px0 - a0, nd = (sh + rw + 3) >> 2;
    // (a cell's ROI lies inside the detection window, i.e. inside the image: level 0 read in place needs nothing but its own base and pitch)
    const bool ip = level == 0 && l0.vbase != nullptr;
    const int pitch = ip ? l0.pitch : g.pitch;
    const uint8_t* src = (ip ? l0.vbase + f * l0.frame_stride : pyr + f * pyr_block + g.plane_off) + (int64_t)py0 * pitch + a0;
    {
      // dword k * 64 + lane of the ROI's rh x nd dwords, eight loads in flight at a time (the address of a dword past the end is
      // clamped to the last one: unconditional loads, so that the compiler counts them instead of waiting for each)
      const int ndw = rh * nd;
      const float rcp_nd = 1.0f / (float)nd;
      for (int k0 = 0; k0 < ndw; k0 += 8 * 64) {
        uint32_t w[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = min(k0 + u * 64 + lane, ndw - 1);
          const int r = (int)(((float)idx + 0.5f) * rcp_nd), d = idx - r * nd;
          dst[u] = r * FC_ND + d;
          w[u] = *reinterpret_cast<const uint32_t*>(src + (int64_t)r * pitch + 4 * d);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u * 64 + lane < ndw) tile32[dst[u]] = w[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- scores at threshold 7: screen with four opposite ring pairs, queue the pixels that pass, exact test on full batches ----
    const float rcp_iw = 1.0f / (float)iw;
    int qn = 0, nc = 0;  // queue length; corners found (their positions remembered while they fit)
    auto score_batch = [&](int firstq, int count) {
      bool corner = false;
      uint32_t pos = 0;
      if (lane < count) {
        const uint32_t e = q[firstq + lane];
        const int best = ring_strength<FC_TPITCH>(tile8 + (e & 0x1fffu));
        pos = e >> 13;  // ix | iy << 6
        corner = best > 7;
        if (corner) score8[((pos >> 6) + 1) * FC_SPITCH + (pos & 0x3f) + 1] = (uint8_t)(best - 1);
      }
      const uint64_t m = ballot64(corner);
      if (m) {
        const int at = nc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (corner && at < FC_CCAP) clist[at] = (uint16_t)pos;
        nc += (int)__popcll(m);
      }
    };
    for (int p0 = 0; p0 < npix; p0 += 64) {
      const int p = p0 + lane;
      bool pass = false;
      uint32_t entry = 0;
      if (p < npix) {
        const int iy = (int)(((float)p + 0.5f) * rcp_iw), ix = p - iy * iw;
        const int off = iy * FC_TPITCH + sh + ix;  // (pixel - 3 rows - 3 columns)
        const uint8_t* rm3 = tile8 + off;
        const int v = rm3[3 * FC_TPITCH + 3];
        const int a0p = rm3[6 * FC_TPITCH + 3], a8 = rm3[3], a4 = rm3[3 * FC_TPITCH + 6], a12 = rm3[3 * FC_TPITCH];
        const int a2 = rm3[5 * FC_TPITCH + 5], a10 = rm3[FC_TPITCH + 1], a6 = rm3[FC_TPITCH + 5], a14 = rm3[5 * FC_TPITCH + 1];
        const int mx = min(min(max(a0p, a8), max(a4, a12)), min(max(a2, a10), max(a6, a14)));
        const int mn = max(max(min(a0p, a8), min(a4, a12)), max(min(a2, a10), min(a6, a14)));
        pass = mx > v + 7 || mn < v - 7;
        entry = (uint32_t)off | ((uint32_t)ix << 13) | ((uint32_t)iy << 19);
      }
      const uint64_t m = ballot64(pass);
      if (m) {
        if (pass) q[qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = entry;
        qn += (int)__popcll(m);
        if (qn >= 64) {
          qn -= 64;
          __builtin_amdgcn_wave_barrier();
          score_batch(qn, 64);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (qn > 0) score_batch(0, qn);
    __builtin_amdgcn_wave_barrier();
    // ---- 3x3 suppression inside the interior (neighbours outside it sit in the zero ring); survivors are compacted into the dead ROI
    // tile as x | y << 12 | score << 24, coordinates relative to (minBorder, minBorder) as the candidate array wants them ----
    int nk = 0;
    auto nms_at = [&](bool valid, int ix, int iy) {
      bool keep = false;
      uint32_t out = 0;
      if (valid) {
        const uint8_t* sc = score8 + (iy + 1) * FC_SPITCH + ix + 1;
        const int ss = sc[0];
        const int nb = max(max(max((int)sc[-FC_SPITCH - 1], (int)sc[-FC_SPITCH]), max((int)sc[-FC_SPITCH + 1], (int)sc[-1])),
                           max(max((int)sc[1], (int)sc[FC_SPITCH - 1]), max((int)sc[FC_SPITCH], (int)sc[FC_SPITCH + 1])));
        keep = ss > nb;  // a score is >= 7: zero (no corner) never passes
        out = (uint32_t)(ox + 3 + ix) | ((uint32_t)(oy + 3 + iy) << 12) | ((uint32_t)ss << 24);
      }
      const uint64_t m = ballot64(keep);
      if (m) {
        if (keep) tile32[nk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = out;
        nk += (int)__popcll(m);
      }
    };
    if (nc <= FC_CCAP) {  // the usual case: only the remembered corners are looked at, and wiped afterwards
      for (int i0 = 0; i0 < nc; i0 += 64) {
        const bool valid = i0 + lane < nc;
        const int pos = valid ? (int)clist[i0 + lane] : 0;
        nms_at(valid, pos & 0x3f, pos >> 6);
      }
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < nc; i += 64) {
        const int pos = clist[i];
        score8[((pos >> 6) + 1) * FC_SPITCH + (pos & 0x3f) + 1] = 0;
      }
    } else {
      for (int p0 = 0; p0 < npix; p0 += 64) {
        const int p = p0 + lane;
        const int iy = (int)(((float)p + 0.5f) * rcp_iw), ix = p - iy * iw;
        nms_at(p < npix, ix, iy);
      }
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < FC_SROWS * FC_SPITCH / 4; i += 64) score32[i] = 0u;
    }
    __builtin_amdgcn_wave_barrier();
    if (nk > 0) {
      int off = 0;
      if (lane == 0) off = atomicAdd(&cursor[2 * ((int64_t)f * L.nlevels + level)], nk);
      off = __builtin_amdgcn_readfirstlane(off);
      const int64_t co = (int64_t)f * cand_block + g.cand_off;
      for (int i = lane; i < nk; i += 64) {
        const uint32_t e = tile32[i];
        if (off + i < g.cand_cap) cand_xy[co + off + i] = (e & 0xfffu) | (((e >> 12) & 0xfffu) << 16), cand_sc[co + off + i] = e >> 24;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Rows per (strip, segment) work item.  A wavefront lives for the whole item, so long segments leave a long under-filled
// tail at the end of the launch (96 rows: +20 % kernel time over 24..32), and the NMS tile of a region has to fit its LDS.
// A single frame has only ~160 items of 24 rows for 1024 SIMDs: short segments spread it over more wavefronts and shorten the launch.  A
// wavefront streams its rows + 8 in blocks of 7: 6 rows of work are exactly two blocks (8 rows take three: 20 -> 16 us for one frame).
#ifndef UVO_FAST_ROWS_FEW
#define UVO_FAST_ROWS_FEW 6
#endif
int fast_rows_per_seg(int batch) { return batch <= 2 ? UVO_FAST_ROWS_FEW : FS_ROWS_MAX; }
int fast_items_per_frame(const Geom& g, int rows_per_seg) {
  int items = 0;
  for (int l = 0; l < g.nlevels; ++l) {
    FastLevel F;
    fast_strip_plan(g.lv[l].w - 32, g.lv[l].h - 32, rows_per_seg, F);
    items += F.items;
  }
  return items;
}
int fast_flags_per_frame(const Geom& g) {
  int n = 0;
  for (int l = 0; l < g.nlevels; ++l) n += g.lv[l].nRows * g.lv[l].nCols;
  return n;
}

FastLevels fast_levels(const Geom& g, int batch) {
  FastLevels L;
  L.nlevels = g.nlevels;
  L.rows_per_seg = fast_rows_per_seg(batch);
  L.items_per_frame = fast_items_per_frame(g, L.rows_per_seg);
  L.flags_per_frame = fast_flags_per_frame(g);
  int fb = 0, first = 0;
  for (int l = 0; l < kMaxLevels; ++l) {
    FastLevel& F = L.l[l];
    if (l < g.nlevels) {
      const LevelGeom& G = g.lv[l];
      F.plane_off = G.plane_off, F.cand_off = G.cand_off, F.pitch = G.pitch, F.cand_cap = G.cand_cap;
      F.w = G.w, F.h = G.h, F.bw = G.bw, F.bh = G.bh, F.nCols = G.nCols, F.nRows = G.nRows, F.wCell = G.wCell, F.hCell = G.hCell;
      F.flag_base = fb;
      F.inv_wcell = (uint32_t)(((1u << 24) + G.wCell - 1) / G.wCell), F.inv_hcell = (uint32_t)(((1u << 24) + G.hCell - 1) / G.hCell);
      fb += G.nRows * G.nCols;
      fast_strip_plan(G.w - 32, G.h - 32, L.rows_per_seg, F);
      F.first_item = first;
      first += F.items;
    } else {
      F = FastLevel{};
      F.w = F.h = 32;
    }
    F.pad = 0;
  }
  return L;
}

// scores + in-cell NMS per region; survivors to the candidate array / the low list of their (frame, level) (the per-cell vote: octree.hip).
// d_tpass[level] = threshold of the level's streaming pass (the lane's adaptive state, see k_octree)
void launch_fast_score(hipStream_t s, const uint8_t* d_pyr, int64_t pyr_block, const Geom& g, int fast_th, const int32_t* d_tpass, uint32_t* d_cor,
                       uint8_t* d_cell_hi, uint32_t* d_cand_xy, uint32_t* d_cand_sc, uint32_t* d_cand_lo, int64_t cand_block, int32_t* d_cursor, int batch,
                       Level0View l0) {
  const FastLevels L = fast_levels(g, batch);
  const dim3 grid((L.items_per_frame + UVO_FAST_WAVES - 1) / UVO_FAST_WAVES, batch);
  hipLaunchKernelGGL(k_fast_score, grid, dim3(64 * UVO_FAST_WAVES), 0, s, d_pyr, pyr_block, L, d_tpass, fast_th, d_cor, d_cell_hi, d_cand_xy, d_cand_sc, d_cand_lo,
                     cand_block, d_cursor, l0);
}

// the sparse second pass at the literal 7 over the cells of threshold-adaptive levels that own no survivor (only needed when fastTh > 7):
// list them, then redo them with a fixed grid of wavefronts (most of which leave at once on textured frames)
void launch_fast_cells(hipStream_t s, const uint8_t* d_pyr, int64_t pyr_block, const Geom& g, const CellDesc* d_cells, const int32_t* d_flag_cell,
                       const int32_t* d_tpass, const uint8_t* d_cell_hi, uint2* d_list, int32_t* d_n_list, uint32_t* d_cand_xy, uint32_t* d_cand_sc,
                       int64_t cand_block, int32_t* d_cursor, int batch, Level0View l0) {
  const FastLevels L = fast_levels(g, batch);
  int max_roi = 0;
  for (int l = 0; l < g.nlevels; ++l) max_roi = std::max(max_roi, std::max(g.lv[l].wCell, g.lv[l].hCell) + 6);
  const bool small = max_roi <= 48;
  if (batch <= UVO_FC_DIRECT_MAX) {  // a latency call: one launch, a wavefront per flag entry
    const dim3 grid((L.flags_per_frame + FC_WAVES - 1) / FC_WAVES, batch);
    if (small)
      hipLaunchKernelGGL((k_fast_cells<48, true>), grid, dim3(64 * FC_WAVES), 0, s, d_pyr, pyr_block, L, d_cells, d_list, d_n_list, d_flag_cell, d_tpass, d_cell_hi, d_cand_xy,
                         d_cand_sc, cand_block, d_cursor, l0);
    else
      hipLaunchKernelGGL((k_fast_cells<66, true>), grid, dim3(64 * FC_WAVES), 0, s, d_pyr, pyr_block, L, d_cells, d_list, d_n_list, d_flag_cell, d_tpass, d_cell_hi, d_cand_xy,
                         d_cand_sc, cand_block, d_cursor, l0);
    return;
  }
  hipLaunchKernelGGL(k_fast_cells_list, dim3((L.flags_per_frame + FCL_THREADS - 1) / FCL_THREADS, batch), dim3(FCL_THREADS), 0, s, d_flag_cell, L.flags_per_frame,
                     g.nlevels, d_tpass, d_cell_hi, d_list, d_n_list, (int)std::min<int64_t>((int64_t)g.total_cells * batch, INT32_MAX));
  // a fixed grid that fills the chip once (seven / three workgroups of four wavefronts per CU: LDS); fewer when the batch cannot hold that many cells
  const int64_t max_items = (int64_t)g.total_cells * batch;
  const int waves = (int)std::min<int64_t>(max_items, 256 * (small ? 7 : 3) * FC_WAVES);
  const dim3 grid((waves + FC_WAVES - 1) / FC_WAVES);
  if (small)
    hipLaunchKernelGGL((k_fast_cells<48, false>), grid, dim3(64 * FC_WAVES), 0, s, d_pyr, pyr_block, L, d_cells, d_list, d_n_list, d_flag_cell, d_tpass, d_cell_hi, d_cand_xy,
                       d_cand_sc, cand_block, d_cursor, l0);
  else
    hipLaunchKernelGGL((k_fast_cells<66, false>), grid, dim3(64 * FC_WAVES), 0, s, d_pyr, pyr_block, L, d_cells, d_list, d_n_list, d_flag_cell, d_tpass, d_cell_hi, d_cand_xy,
                       d_cand_sc, cand_block, d_cursor, l0);
}

}  // namespace uvo
