// Host check of the strip plan (u-vip-slam_amd/csrc/strip_plan.hpp): for a window of w x h pixels and a segment height, every pixel
// must be owned by exactly one (item, sub-strip, lane group), items must be decodable, and no strip may need more lanes than a
// wavefront has.  Returns 0 when the plan is sound, a positive code naming the violated property otherwise.
#include <cstddef>
#include <vector>
using std::size_t;

#include "../../u-vip-slam_amd/csrc/strip_plan.hpp"

extern "C" int emu_strip_plan_check(int w, int h, int rows_per_seg, int* items_out, int* lane_rows_out) {
  using namespace uvo;
  StripPlan P;
  fast_strip_plan(w, h, rows_per_seg, P);
  *items_out = P.items;
  std::vector<int> cover((size_t)w * h, 0);
  long lane_rows = 0;
  for (int item = 0; item < P.items; ++item) {
    int strip_x, seg, sub;
    fast_strip_item(P, item, strip_x, seg, sub);
    if (sub != 1 && sub != 2 && sub != 4) return 1;
    const int cols = fast_sub_cols(sub);          // columns a sub-strip owns
    if (cols != 256 / sub - 8 || cols < 1) return 2;
    if (strip_x < 0 || strip_x >= w) return 3;    // a strip must start inside the window
    lane_rows += rows_per_seg;
    for (int s = 0; s < sub; ++s) {
      const int y0 = (seg + s) * rows_per_seg;
      if (y0 >= h) continue;                      // sub-strip below the window: idle lanes
      const int y1 = y0 + rows_per_seg < h ? y0 + rows_per_seg : h;
      for (int y = y0; y < y1; ++y)
        for (int x = strip_x; x < strip_x + cols && x < w; ++x) ++cover[(size_t)y * w + x];
    }
  }
  for (size_t i = 0; i < cover.size(); ++i)
    if (cover[i] != 1) return 10 + (cover[i] > 1);
  *lane_rows_out = (int)lane_rows;
  return 0;
}

// xcd_contiguous must be a permutation of [0, total) that keeps the workgroups of one XCD (b % 8) in one contiguous range
extern "C" int emu_xcd_contiguous_check(int total) {
  std::vector<int> seen(total, 0);
  std::vector<int> lo(8, total), hi(8, -1), cnt(8, 0);
  for (int b = 0; b < total; ++b) {
    const int v = uvo::xcd_contiguous(b, total);
    if (v < 0 || v >= total) return 1;
    if (seen[v]++) return 2;
    const int x = b & 7;
    lo[x] = v < lo[x] ? v : lo[x], hi[x] = v > hi[x] ? v : hi[x], ++cnt[x];
  }
  for (int x = 0; x < 8; ++x)
    if (cnt[x] && hi[x] - lo[x] + 1 != cnt[x]) return 3;  // not contiguous
  return 0;
}
