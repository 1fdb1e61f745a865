// Strip plan shared by the streaming stencil kernels (k_fast_score, k_gauss7): how a level's window is cut into wavefront strips.
// Free of HIP headers so that the host-side tests can compile it with a plain C++ compiler (tests/emu/strip_plan_emu.cpp).
#pragma once
#if defined(__HIPCC__)
#define UVO_PLAN_HD __host__ __device__
#else
#define UVO_PLAN_HD
#endif

namespace uvo {

constexpr int FS_COLS = 248;  // useful columns per wavefront strip (lanes 1..62 of 64 lanes x 4 pixels)

// Strip plan of one level.  A wavefront is 64 lanes x 4 pixels wide; a level's window is rarely a multiple of the 248 useful
// columns, and a wavefront that owns a 5-pixel remainder costs as much as a full one.  So the remainder is cut into narrow strips
// of at most 120 (two segments side by side, 32 lanes each) or 56 columns (four segments, 16 lanes each); a remainder wider than
// 176 columns stays one ordinary strip.  Columns owned by a sub-strip of L lanes: 4 L - 8 (its first and last lane are halo).
UVO_PLAN_HD inline int fast_sub_cols(int sub) { return 256 / sub - 8; }
template <class PlanT>
UVO_PLAN_HD inline void fast_strip_plan(int window_w, int window_h, int rows_per_seg, PlanT& F) {
  F.nseg = (window_h + rows_per_seg - 1) / rows_per_seg;
  F.nfull = window_w / FS_COLS;
  int rem = window_w - F.nfull * FS_COLS, x = F.nfull * FS_COLS;
  F.sub[0] = F.sub[1] = 0, F.x0[0] = F.x0[1] = 0;
  if (rem > fast_sub_cols(2) + fast_sub_cols(4)) {
    F.nfull += 1;
    rem = 0;
  }
  for (int k = 0; k < 2 && rem > 0; ++k) {
    F.sub[k] = rem > fast_sub_cols(4) ? 2 : 4;
    F.x0[k] = x;
    x += fast_sub_cols(F.sub[k]);
    rem -= fast_sub_cols(F.sub[k]);
  }
  int narrow = 0;
  for (int k = 0; k < 2; ++k)
    if (F.sub[k]) narrow += (F.nseg + F.sub[k] - 1) / F.sub[k];
  if (narrow >= F.nseg && (F.sub[0] || F.sub[1])) {  // few segments: grouping them saves nothing, one ordinary strip is as cheap
    F.nfull += 1;
    F.sub[0] = F.sub[1] = 0, F.x0[0] = F.x0[1] = 0;
    narrow = 0;
  }
  F.items = F.nfull * F.nseg + narrow;
}

// item of a planned level -> (first window column of the strip, first segment, sub-strips)
template <class PlanT>
UVO_PLAN_HD inline void fast_strip_item(const PlanT& F, int item, int& strip_x, int& seg, int& sub) {
  if (item < F.nfull * F.nseg) {
    strip_x = (item % F.nfull) * FS_COLS, seg = item / F.nfull, sub = 1;
  } else {
    item -= F.nfull * F.nseg;
    const int n0 = F.sub[0] ? (F.nseg + F.sub[0] - 1) >> (F.sub[0] >> 1) : 0;  // sub is 2 or 4: a shift by 1 or 2, not a division
    const int k = item < n0 ? 0 : 1;
    if (k) item -= n0;
    sub = F.sub[k], strip_x = F.x0[k], seg = item * sub;
  }
}
struct StripPlan {
  int nfull, nseg, items;
  int sub[2], x0[2];
};


// Workgroups are dealt round-robin over the 8 XCDs (workgroups b and b + 8 share an XCD and its L2).  This maps the linear
// workgroup index to a virtual one such that each XCD walks a contiguous range of virtual indices: neighbouring work items
// (strips / segments that re-read each other's halo rows) then meet in one L2.  Speed only, never correctness.
UVO_PLAN_HD inline int xcd_contiguous(int b, int total) {
  const int per = total >> 3, rem = total & 7;  // the first `rem` XCDs own per + 1 workgroups
  const int x = b & 7, i = b >> 3;
  return x < rem ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
}

}  // namespace uvo
