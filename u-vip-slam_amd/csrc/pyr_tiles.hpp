// Tile plan of the fused pyramid launch (k_pyr_tiles, pyramid.hip): ComputePyramid (src/ORBextractor.cc:963-1004) as ONE launch.
//
// The chain of the reference -- level l is cv::resize of level l - 1 (:982) -- is a chain per PIXEL NEIGHBOURHOOD, not per level: a tile of
// level l needs the tile of level l - 1 under it and a pixel or two around that.  A workgroup owns one cell of a TX x TY partition of
// every level of a GROUP of consecutive levels first .. last (what it stores to HBM) and walks them in turn; level l is computed from level
// l - 1's tile in LDS (the group's first level: from memory -- the image, or the plane the previous group's launch stored), over the
// workgroup's own cell AND the few pixels around it that its cell of level l + 1 will read -- those are computed again by the neighbouring
// workgroup that owns them (same table entries, same integer arithmetic: the same bytes), which is what makes the tiles independent: one
// launch per group, one workgroup barrier per level, no level of a group is read back from memory.  The halo grows by about 1.2 x + 4
// pixels per level below the group's last one, so deep groups pay in redundant pixels (latency path: one or two groups of many small
// tiles; throughput path: shallow groups of large tiles).
//
// Everything here is host code free of HIP headers (tests/emu/pyr_tiles_emu.cpp executes the plan on the CPU against the oracle's planes).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace uvo {

// resize coefficient tables, indexed by padded output coordinates (border reflection folded in); built by extractor.cpp
struct ResizeCol {  // 8 B: left tap column, the two 11-bit weights scaled by 16 (a << 4 <= 32768); pad: window base / v_perm selector halves
  uint16_t sx, a0, a1, pad;
};
struct ResizeRow {  // 8 B
  int16_t sy0, sy1, b0, b1;
};

constexpr int kPyrTilePad = 16;    // EDGE_THRESHOLD: ROI origin inside a padded plane
constexpr int kPyrTileRows = 4;    // output rows per work item (a row group of the row table)
constexpr int kPyrTileSlack = 12;  // bytes behind an LDS tile row that a 12-byte tap window may touch (taps of weight zero)

// cv::resize INTER_LINEAR coefficient tables exactly as resizeGeneric_ builds them:
// fx = (float)((dx+0.5)*scale_x - 0.5), sx = floor(fx), weights saturate_cast<short>(w * 2048); then re-indexed by
// padded output coordinate with the REFLECT_101 border folded in (copyMakeBorder of the level, src/ORBextractor.cc:988).
inline int pyr_round_host(float v) { return (int)lrintf(v); }  // cvRound: half to even
inline int pyr_floor_host(float v) {
  int i = (int)v;
  return i - (i > v);
}
inline int pyr_reflect101_host(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}
// Tables of level l (l >= 1) from level l - 1: `pitch` column entries, (ph + 3) & ~3 row entries (ph = dh + 32).
// fast_ok: every output dword of the level finds its eight taps inside one 12-byte aligned source window (true for scale factors up to
// ~1.33); otherwise the level takes the byte-gather path of k_resize_level.
inline void pyr_build_level_tables(int sw, int sh, int dw, int dh, int pitch, std::vector<ResizeCol>& ctab, std::vector<ResizeRow>& rtab, int* fast_ok) {
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  const int ph = dh + 2 * kPyrTilePad;
  std::vector<ResizeCol> col(dw);
  std::vector<ResizeRow> row(dh);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = pyr_floor_host(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= sw - 1) fx = 0, sx = sw - 1;
    col[dx] = ResizeCol{(uint16_t)sx, (uint16_t)(pyr_round_host((1.f - fx) * 2048.f) << 4), (uint16_t)(pyr_round_host(fx * 2048.f) << 4), 0};
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = pyr_floor_host(fy);
    fy -= sy;
    const int sy0 = std::min(std::max(sy, 0), sh - 1), sy1 = std::min(std::max(sy + 1, 0), sh - 1);
    row[dy] = ResizeRow{(int16_t)sy0, (int16_t)sy1, (int16_t)pyr_round_host((1.f - fy) * 2048.f), (int16_t)pyr_round_host(fy * 2048.f)};
  }
  *fast_ok = 1;
  for (int px = 0; px < pitch; px += 4) {
    ResizeCol e[4];
    int lo = 1 << 30;
    for (int i = 0; i < 4; ++i) {
      e[i] = col[pyr_reflect101_host(px + i - kPyrTilePad, dw)];  // columns in the pitch slack map to something valid too
      lo = std::min(lo, (int)e[i].sx);
    }
    // per dword: window base (multiple of 4) and the v_perm selector = offsets of the four left taps inside the window
    const int base = lo & ~3;
    uint32_t sel = 0;
    for (int i = 0; i < 4; ++i) {
      const int o = (int)e[i].sx - base;
      if (o > 7) *fast_ok = 0;
      sel |= (uint32_t)(o & 0xff) << (8 * i);
    }
    e[0].pad = (uint16_t)base, e[1].pad = (uint16_t)(sel & 0xffff), e[2].pad = (uint16_t)(sel >> 16), e[3].pad = 0;
    for (int i = 0; i < 4; ++i) ctab.push_back(e[i]);
  }
  for (int py = 0; py < ((ph + 3) & ~3); ++py) rtab.push_back(row[pyr_reflect101_host(std::min(py, ph - 1) - kPyrTilePad, dh)]);  // padded to groups of 4
}

// What a workgroup does on one level (40 B).  Coordinates: "padded" = the level's padded plane (ROI origin at (16, 16)); "ROI" = the level's image.
struct PyrTileLevel {
  int16_t cx0w, ncw;   // computed region: first dword column (padded x / 4) and number of dword columns
  int16_t cy0, nrg;    // first padded row (multiple of 4) and number of row groups of 4
  int16_t ox0w, ox1w;  // owned (stored to HBM) dword columns [ox0w, ox1w) ...
  int16_t oy0, oy1;    // ... and padded rows [oy0, oy1)
  int16_t lx0, ly0;    // LDS tile of this level (what level + 1 reads): ROI coordinates of its first byte, lx0 % 4 == 0
  int16_t lw, lrows;   // bytes per tile row that are stored (multiple of 4; the row pitch is lw + kPyrTileSlack) and rows; 0 x 0: no tile (last level)
  uint32_t ncw_magic;  // ceil(2^32 / ncw)
  uint32_t lds_off;    // byte offset of the tile inside the workgroup's LDS
  uint32_t tab_off;    // byte offset of the level's coefficient tables in LDS (levels above the group's first): ncw x 2 uint4 of column entries
                       // (dword columns cx0w ..), then nrg x 2 uint4 of row entries (row groups cy0 / 4 ..)
  uint32_t pad;
};

struct PyrTilePlan {
  int tx = 0, ty = 0, nlevels = 0, first = 0, last = 0;  // the group's levels first .. last (1 <= first <= last < nlevels)
  uint32_t lds_bytes = 0;
  std::vector<PyrTileLevel> lv;  // [tile][level]; entry of level 0 unused
  int64_t computed_px = 0, owned_px = 0;  // per frame: the redundancy of the plan = computed / owned
};

struct PyrTileDims {
  int w, h, pitch;  // ROI size, row pitch of the padded plane
};

// cell boundaries of a partition of [lo, hi) into n cells, aligned to `align` (first = lo, last = hi; lo and hi are multiples of 4)
inline void pyr_tile_cuts(int lo, int hi, int n, int align, std::vector<int>& cuts) {
  cuts.assign(n + 1, lo);
  for (int k = 1; k < n; ++k) {
    int c = lo + (int)((int64_t)(hi - lo) * k / n);
    c = (c + align / 2) / align * align;
    cuts[k] = std::min(std::max(c, cuts[k - 1]), hi);
  }
  cuts[n] = hi;
}

// Builds the plan of the group of levels first .. last.  dims[l], ctab[l] (one entry per padded column, `pitch` entries), rtab[l] (one entry per padded row, padded to whole groups
// of 4) for l = 1 .. nlevels - 1; ring = border pixels written around the ROI (4: all that is ever read).  Returns false when a level does not
// take the 12-byte-window path or a tile does not fit `max_lds` bytes.
inline bool pyr_tile_plan_build(const PyrTileDims* dims, int nlevels, int first, int last, const ResizeCol* const* ctab, const ResizeRow* const* rtab, int ring, int tx,
                                int ty, uint32_t max_lds, PyrTilePlan& P) {
  P = PyrTilePlan();
  P.tx = tx, P.ty = ty, P.nlevels = nlevels, P.first = first, P.last = last;
  if (nlevels < 2 || first < 1 || last < first || last >= nlevels || tx < 1 || ty < 1 || ring < 0 || ring > kPyrTilePad || ring % 4) return false;
  const int ntiles = tx * ty;
  P.lv.assign((size_t)ntiles * nlevels, PyrTileLevel{});
  // the written area of a level: padded columns [16 - ring, round4(16 + w + ring)) and rows [16 - ring, 16 + h + ring), as the per-level launches write it
  std::vector<std::vector<int>> xc(nlevels), yc(nlevels);
  for (int l = first; l <= last; ++l) {
    const int x_lo = kPyrTilePad - ring, x_hi = (kPyrTilePad + dims[l].w + ring + 3) / 4 * 4;
    const int y_lo = kPyrTilePad - ring, y_hi = kPyrTilePad + dims[l].h + ring;
    if (x_hi > dims[l].pitch) return false;
    const int cw = (x_hi - x_lo) / tx;
    pyr_tile_cuts(x_lo, x_hi, tx, cw >= 96 ? 32 : (cw >= 40 ? 16 : 4), xc[l]);
    pyr_tile_cuts(y_lo, (y_hi + 3) / 4 * 4, ty, 4, yc[l]);
    yc[l][ty] = y_hi;  // the last cell ends with the plane's last written row (its last row group is partial)
  }
  uint32_t buf_bytes[2] = {0, 0};
  for (int t = 0; t < ntiles; ++t) {
    const int ti = t % tx, tj = t / tx;
    // the group's last level down: what level l computes is its own cell plus what level l + 1's computed region reads
    int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;  // need of the level above, ROI coordinates of THIS level; empty for the top level
    bool have_need = false;
    for (int l = last; l >= first; --l) {
      PyrTileLevel& T = P.lv[(size_t)t * nlevels + l];
      int ox0 = xc[l][ti], ox1 = xc[l][ti + 1], oy0 = yc[l][tj], oy1 = yc[l][tj + 1];
      int cx0 = ox0, cx1 = ox1, cy0 = oy0, cy1 = oy1;
      if (ox1 <= ox0 || oy1 <= oy0) cx0 = cx1 = ox0 = ox1 = 0, cy0 = cy1 = oy0 = oy1 = 0;  // an empty cell (more tiles than pixels)
      if (have_need) {
        const int px0 = (nx0 + kPyrTilePad) & ~3, px1 = (nx1 + kPyrTilePad + 3) & ~3, py0 = ny0 + kPyrTilePad, py1 = ny1 + kPyrTilePad;
        if (cx1 <= cx0) cx0 = px0, cx1 = px1, cy0 = py0, cy1 = py1;
        else cx0 = std::min(cx0, px0), cx1 = std::max(cx1, px1), cy0 = std::min(cy0, py0), cy1 = std::max(cy1, py1);
        T.lx0 = (int16_t)(nx0 & ~3), T.ly0 = (int16_t)ny0;
        T.lw = (int16_t)(((nx1 + 3) & ~3) - T.lx0), T.lrows = (int16_t)(ny1 - ny0);
      }
      cy0 &= ~3;
      const int nrg = (cy1 - cy0 + kPyrTileRows - 1) / kPyrTileRows;
      T.cx0w = (int16_t)(cx0 / 4), T.ncw = (int16_t)((cx1 - cx0) / 4), T.cy0 = (int16_t)cy0, T.nrg = (int16_t)nrg;
      T.ox0w = (int16_t)(ox0 / 4), T.ox1w = (int16_t)(ox1 / 4), T.oy0 = (int16_t)oy0, T.oy1 = (int16_t)oy1;
      T.ncw_magic = T.ncw > 0 ? (uint32_t)((0x100000000ull + T.ncw - 1) / T.ncw) : 0u;
      P.computed_px += (int64_t)(cx1 - cx0) * nrg * kPyrTileRows, P.owned_px += (int64_t)(ox1 - ox0) * (oy1 - oy0);
      if (T.lw > 0) {
        const uint32_t bytes = (uint32_t)(T.lw + kPyrTileSlack) * (uint32_t)T.lrows;
        buf_bytes[l & 1] = std::max(buf_bytes[l & 1], (bytes + 15u) & ~15u);
      }
      // what this level's computed region reads of level l - 1 (every row of every row group is evaluated, also the ones past the region's end)
      have_need = false;
      if (T.ncw > 0 && nrg > 0) {
        const int sw = dims[l - 1].w;
        nx0 = 1 << 30, nx1 = -1, ny0 = 1 << 30, ny1 = -1;
        for (int cw = T.cx0w; cw < T.cx0w + T.ncw; ++cw) {
          const ResizeCol* e = ctab[l] + 4 * cw;
          const int base = e[0].pad;
          for (int i = 0; i < 4; ++i) {
            const int o = (int)e[i].sx - base;
            if (o < 0 || o > 7) return false;  // the level does not take the 12-byte window
            nx0 = std::min(nx0, base), nx1 = std::max(nx1, std::min((int)e[i].sx + 1, sw - 1) + 1);
          }
        }
        for (int py = cy0; py < cy0 + nrg * kPyrTileRows; ++py) {
          const ResizeRow& r = rtab[l][py];
          ny0 = std::min(ny0, (int)std::min(r.sy0, r.sy1)), ny1 = std::max(ny1, (int)std::max(r.sy0, r.sy1) + 1);
        }
        have_need = true;
      }
    }
  }
  // levels alternate between two LDS regions: level l writes region l & 1 while it reads region (l - 1) & 1; behind them the coefficient
  // tables of the levels above the first (staged once at the start of the launch: nothing but the first level's taps waits for memory)
  uint32_t tab_bytes = 0;
  for (int t = 0; t < ntiles; ++t) {
    uint32_t o = buf_bytes[0] + buf_bytes[1];
    for (int l = first; l <= last; ++l) {
      PyrTileLevel& T = P.lv[(size_t)t * nlevels + l];
      T.lds_off = (l & 1) ? 0u : buf_bytes[1];
      T.tab_off = o;
      if (l > first) o += (uint32_t)(T.ncw + T.nrg) * 32u;
    }
    tab_bytes = std::max(tab_bytes, o - (buf_bytes[0] + buf_bytes[1]));
  }
  P.lds_bytes = buf_bytes[0] + buf_bytes[1] + tab_bytes;
  return P.lds_bytes <= max_lds;
}

}  // namespace uvo
