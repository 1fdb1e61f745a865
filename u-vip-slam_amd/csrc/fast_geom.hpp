// Geometry of the FAST stage shared by k_fast_score (fast.hip) and the candidate gathering in front of the quad-tree
// (octree.hip): (strip, segment) regions, their corner lists, per-level cell grids.
#pragma once
#include "common.hpp"

namespace uvo {

constexpr int FS_COLS = 248;      // useful columns per wavefront strip (lanes 1..62)
constexpr int FS_ROWS_MAX = 24;   // rows per (strip, segment) region; bounded by the NMS tile that must fit the wavefront's LDS
constexpr int FS_REGION_ENTRIES = (FS_COLS + 2) * (FS_ROWS_MAX + 2);  // corner list capacity: the region plus its halo ring

struct FastLevel {  // per-level values of the sparse stages, passed in the kernel argument block (scalar loads)
  int64_t plane_off, cand_off;
  int pitch, cand_cap;
  int w, h, bw, bh;
  int nCols, nRows, wCell, hCell;
  int flag_base;  // first entry of this level in the per-frame cell-flag array (full nRows x nCols grid)
  int pad;
  uint32_t inv_wcell, inv_hcell;  // ceil(2^32 / wCell), ceil(2^32 / hCell): n / cell = umulhi(n, inv) for the coordinate range
};
struct FastLevels {
  FastLevel l[kMaxLevels];
  int nlevels, rows_per_seg, items_per_frame, flags_per_frame;
};

// region id -> (level, strip, segment); false when the wavefront has no region
#ifdef __HIPCC__
__device__ __forceinline__ bool fast_region(const FastLevels& L, int item, int& level, int& X0, int& py0) {
  for (level = 0; level < L.nlevels; ++level) {
    const int nstrip = (L.l[level].w - 32 + FS_COLS - 1) / FS_COLS;
    const int nseg = (L.l[level].h - 32 + L.rows_per_seg - 1) / L.rows_per_seg;
    if (item < nstrip * nseg) {
      X0 = 28 + (item % nstrip) * FS_COLS;
      py0 = 32 + (item / nstrip) * L.rows_per_seg;
      return true;
    }
    item -= nstrip * nseg;
  }
  return false;
}
#endif

int fast_rows_per_seg(int batch);
int fast_items_per_frame(const Geom& g, int rows_per_seg);
int fast_flags_per_frame(const Geom& g);
FastLevels fast_levels(const Geom& g, int batch);

}  // namespace uvo
