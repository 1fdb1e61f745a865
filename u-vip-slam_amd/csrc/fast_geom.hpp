// Geometry of the FAST stage shared by k_fast_score (fast.hip) and the candidate gathering in front of the quad-tree
// (octree.hip): (strip, segment) regions, their corner lists, per-level cell grids.
#pragma once
#include "common.hpp"
#include "strip_plan.hpp"

namespace uvo {

#ifndef UVO_FAST_ROWS
#define UVO_FAST_ROWS 24
#endif
constexpr int FS_ROWS_MAX = UVO_FAST_ROWS;   // rows per (strip, segment) region; bounded by the NMS tile that must fit the wavefront's LDS (and by 30: the row mask is a dword)
constexpr int FS_REGION_ENTRIES = (FS_COLS + 2) * (FS_ROWS_MAX + 2);  // corner list capacity: the region plus its halo ring

struct FastLevel {  // per-level values of the sparse stages, passed in the kernel argument block (scalar loads)
  int64_t plane_off, cand_off;
  int pitch, cand_cap;
  int w, h, bw, bh;
  int nCols, nRows, wCell, hCell;
  int flag_base;  // first entry of this level in the per-frame cell-flag array (full nRows x nCols grid)
  int pad;
  uint32_t inv_wcell, inv_hcell;  // ceil(2^24 / wCell), ceil(2^24 / hCell): n / cell = (n * inv) >> 24 in 24-bit multiplies, exact for n < 4096
                                  // (n * (inv * cell - 2^24) < 2^24: the excess is below the cell size, at most 66)
  // strips of the detection window (see fast_strip_plan): nfull wavefront-wide strips of FS_COLS columns, then up to two narrow ones
  // in which a wavefront walks `sub[k]` row segments side by side (32 or 16 lanes each)
  int nfull, nseg, items, first_item;
  int sub[2], x0[2];
};
struct FastLevels {
  FastLevel l[kMaxLevels];
  int nlevels, rows_per_seg, items_per_frame, flags_per_frame;
};

// region id -> (level, first padded column, first padded row of sub-strip 0, sub-strips); false when the wavefront has no region
#ifdef __HIPCC__
__device__ __forceinline__ bool fast_region(const FastLevels& L, int item, int& level, int& X0, int& py0, int& sub) {
  for (level = 0; level < L.nlevels; ++level) {
    const FastLevel& F = L.l[level];
    if (item < F.items) {
      int strip_x, seg;
      fast_strip_item(F, item, strip_x, seg, sub);
      X0 = 28 + strip_x;
      py0 = 32 + seg * L.rows_per_seg;
      return true;
    }
    item -= F.items;
  }
  return false;
}
#endif

int fast_rows_per_seg(int batch);
int fast_items_per_frame(const Geom& g, int rows_per_seg);
int fast_flags_per_frame(const Geom& g);
FastLevels fast_levels(const Geom& g, int batch);

}  // namespace uvo
