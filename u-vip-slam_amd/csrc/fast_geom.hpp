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
  // strips of the detection window (see fast_strip_plan): nfull wavefront-wide strips of FS_COLS columns, then up to two narrow ones
  // in which a wavefront walks `sub[k]` row segments side by side (32 or 16 lanes each)
  int nfull, nseg, items, first_item;
  int sub[2], x0[2];
};
struct FastLevels {
  FastLevel l[kMaxLevels];
  int nlevels, rows_per_seg, items_per_frame, flags_per_frame;
};

// Strip plan of one level.  A wavefront is 64 lanes x 4 pixels wide; a level's window is rarely a multiple of the 248 useful
// columns, and a wavefront that owns a 5-pixel remainder costs as much as a full one.  So the remainder is cut into narrow strips
// of at most 120 (two segments side by side, 32 lanes each) or 56 columns (four segments, 16 lanes each); a remainder wider than
// 176 columns stays one ordinary strip.  Columns owned by a sub-strip of L lanes: 4 L - 8 (its first and last lane are halo).
inline __host__ __device__ int fast_sub_cols(int sub) { return 256 / sub - 8; }
template <class PlanT>
inline __host__ __device__ void fast_strip_plan(int window_w, int window_h, int rows_per_seg, PlanT& F) {
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
  F.items = F.nfull * F.nseg;
  for (int k = 0; k < 2; ++k)
    if (F.sub[k]) F.items += (F.nseg + F.sub[k] - 1) / F.sub[k];
}

// item of a planned level -> (first window column of the strip, first segment, sub-strips)
template <class PlanT>
inline __host__ __device__ void fast_strip_item(const PlanT& F, int item, int& strip_x, int& seg, int& sub) {
  if (item < F.nfull * F.nseg) {
    strip_x = (item % F.nfull) * FS_COLS, seg = item / F.nfull, sub = 1;
  } else {
    item -= F.nfull * F.nseg;
    const int n0 = F.sub[0] ? (F.nseg + F.sub[0] - 1) / F.sub[0] : 0;
    const int k = item < n0 ? 0 : 1;
    if (k) item -= n0;
    sub = F.sub[k], strip_x = F.x0[k], seg = item * sub;
  }
}
struct StripPlan {
  int nfull, nseg, items;
  int sub[2], x0[2];
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
