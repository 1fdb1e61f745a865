// Shared host/device declarations of the extractor pipeline: per-level geometry, scratch layout, launch
// wrappers.  Data layout in HBM (see DESIGN.md "Data layout"):
//   pyramid  : [frame][level] padded u8 planes, row pitch = (w+32) rounded up to 64 B, ROI origin at (16,16)
//   blurred  : same geometry, 7x7 sigma-2 blurred interior + un-blurred reflected pad ring
//   corners  : [frame][region][250 x 26] packed corner records, one region per k_fast_score wavefront -- overflow storage only: a
//              wavefront keeps its corner list in LDS and moves it here when a region holds more than 384 corners
//   cand     : [frame][level][cap_l] FAST candidates, SoA words {x | y<<16} and {score}, count in cand_count[frame][level]
//   cand_lo  : [frame][level][cap_l] NMS survivors below fastTh (x | y<<12 | score<<24) waiting for the per-cell threshold vote
//   cursor   : [frame][level][2] fill counts of cand (survivors >= fastTh, and the literal-7 survivors of k_fast_cells) and cand_lo; zero
//              between batches
//   cell_hi  : [frame][sum of nRows x nCols] a cell owns a survivor >= fastTh; tpass / fcount / cell list: the adaptive FAST mode (fast.hip)
//   sel      : [frame][level][quota_l+4] quad-tree survivors in the reference's list order
//   flist    : [frame][flist_cap] final (level, x, y, aux) slots in output order
#pragma once
#include "strip_plan.hpp"
#include "pyr_tiles.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uvo/uvo.h"

namespace uvo {

constexpr int kMaxLevels = 16;
constexpr int kPad = 16;         // EDGE_THRESHOLD, src/ORBextractor.cc:78
constexpr int kPyrPitchAlign = 64;  // row pitch of the padded planes (128 -- whole cache lines -- grew the planes by 7 %: every stage paid, profiles/r04_pyramid_ab.txt)
constexpr int kMinBorder = 13;   // EDGE_THRESHOLD-3, src/ORBextractor.cc:756
constexpr int kMaxOctN = 2000;   // largest per-level quota the quad-tree kernel is sized for (LDS budget)

struct LevelGeom {
  int w, h;            // ROI size
  int pw, ph, pitch;   // padded width/height, row pitch in bytes
  int64_t plane_off;   // byte offset of the padded plane inside a frame's pyramid block
  int bw, bh;          // detection window size: (w-26) x (h-26)   (maxBorder - minBorder)
  int nCols, nRows, wCell, hCell;
  int cell_base;       // index of this level's first cell in the per-frame cell table
  int n_cells;
  int quota;           // mnFeaturesPerLevel
  int cand_cap;        // worst-case FAST survivors
  int64_t cand_off;    // element offset into a frame's candidate block
  int sel_cap;         // quota + 4
  int sel_off;         // element offset into a frame's sel block
  int nIni;            // quad-tree roots
  float hX;
  float scale;         // mvScaleFactor[level]
  float patch_size;    // (float)(int)(31*scale)
  int xtab_off;        // element offset of this level's column resize table (level >= 1; one entry per padded column, pitch entries)
  int ytab_off;        // same for the row table (ph entries)
  int oct_tab_off;     // element offset of this level's quad-tree path tables (bw column entries, then bh row entries; octree_fill_path_tables)
  int pad_;
};

// Level 0 read in place: the caller's image instead of the padded plane (cv::copyMakeBorder at :996 is then never materialised).
// vbase = image - 16 * pitch - 16: the address padded coordinates (row, column) of level 0 would have if the image sat inside a padded
// plane of row pitch `pitch`; only pixels inside the image may be dereferenced (a reader that needs the 16-pixel border reflects the
// index itself).  vbase = NULL: level 0 is the padded plane like every other level.
struct Level0View {
  const uint8_t* vbase;
  int64_t frame_stride;
  int pitch;
  int pad;
};

struct CellDesc {  // one FAST cell (src/ORBextractor.cc:773-790)
  int16_t level;
  int16_t x0, y0;    // ROI origin in level coordinates (iniX, iniY)
  int16_t rw, rh;    // ROI size (maxX-iniX, maxY-iniY)
  int16_t ox, oy;    // j*wCell, i*hCell: shift applied to ROI coords -> coords relative to minBorder
  int16_t pad;
};

struct Geom {
  int width, height, nlevels;
  int total_cells;
  int64_t pyr_block;    // bytes per frame
  int64_t cand_block;   // candidates per frame
  int sel_block;        // sel entries per frame
  int flist_cap;        // final-list slots per frame
  LevelGeom lv[kMaxLevels];
};

// FAST candidates and quad-tree survivors are SoA: xy word = x | y << 16 relative to (minBorder, minBorder),
// score word = FAST score (cornerScore), 1..254.

struct FinalSlot {  // 16 B
  float x, y;       // level coordinates
  int32_t level;    // bit 31 set: caller keypoint, aux = index into in_kp
  int32_t aux;      // FAST score for detected points
};

const char* hip_err_set(hipError_t e, const char* what);
#ifdef __HIPCC__
// index of the calling wavefront inside its workgroup, as a scalar: the compiler cannot prove threadIdx.x >> 6 wave-uniform by
// itself, and everything derived from it (row counters, queue lengths, addresses) would otherwise live in vector registers.
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#endif

int fail(int code, const char* msg);  // records msg for uvo_last_error() and returns code

#define UVO_HIP_CHECK(expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) {                                   \
      uvo::hip_err_set(_e, #expr);                            \
      return UVO_E_HIP;                                       \
    }                                                         \
  } while (0)

// ---- kernel launch wrappers (defined in the .hip files) ----
void launch_pad_level0(hipStream_t s, const uint8_t* d_img, int w, int h, int64_t stride, int64_t frame_stride, uint8_t* d_pyr,
                       int64_t pyr_block, const LevelGeom& g0, int batch);
void launch_resize_level(hipStream_t s, uint8_t* d_pyr, int64_t pyr_block, const LevelGeom& src, const LevelGeom& dst, const ResizeCol* d_ctab,
                         const ResizeRow* d_rtab, int fast_ok, int batch, Level0View l0, int ring);
// a group of consecutive levels in one launch (pyramid.hip: k_pyr_tiles; plan: pyr_tiles.hpp)
constexpr int kPyrTilesMaxLds = 160 * 1024;
int prepare_pyr_tiles(uint32_t* max_lds);  // *max_lds: bytes of LDS a tile plan may use on the current device
int launch_pyr_tiles(hipStream_t s, uint8_t* d_pyr, int64_t pyr_block, const PyrTileLevel* d_plan, const Geom& g, const ResizeCol* d_ctab, const ResizeRow* d_rtab,
                     Level0View l0, int first, int last, int ntiles, uint32_t lds_bytes, int threads, int rows, int batch);
void launch_gauss7(hipStream_t s, const uint8_t* d_pyr, uint8_t* d_blur, int64_t pyr_block, const LevelGeom* d_lv, const Geom& g, int4 taps,
                   int batch, int sse2_rounding, Level0View l0);
void launch_fast_score(hipStream_t s, const uint8_t* d_pyr, int64_t pyr_block, const Geom& g, int fast_th, const int32_t* d_tpass, uint32_t* d_cor,
                       uint8_t* d_cell_hi, uint32_t* d_cand_xy, uint32_t* d_cand_sc, uint32_t* d_cand_lo, int64_t cand_block, int32_t* d_cursor, int batch,
                       Level0View l0);
void launch_fast_cells(hipStream_t s, const uint8_t* d_pyr, int64_t pyr_block, const Geom& g, const CellDesc* d_cells, const int32_t* d_flag_cell,
                       const int32_t* d_tpass, const uint8_t* d_cell_hi, uint2* d_list, int32_t* d_n_list, uint32_t* d_cand_xy, uint32_t* d_cand_sc,
                       int64_t cand_block, int32_t* d_cursor, int batch, Level0View l0);
void launch_grider(hipStream_t s, const uint8_t* d_img, int w, int h, int64_t stride, int num_features, int grid_x, int grid_y, int threshold,
                   int nms, uint8_t* d_score, uint32_t* d_lists, int32_t* d_counts, uvo_keypoint* d_out, int cap, int32_t* d_n_out);
int fast_rows_per_seg(int batch);
int fast_items_per_frame(const Geom& g, int rows_per_seg);
int fast_flags_per_frame(const Geom& g);
// Launch shape of the quad-tree kernel, per extractor handle (nothing process-global: handles on several devices and host
// threads coexist in one process).  wide_max_problems: up to this many (frame, level) problems run as 1024-thread workgroups.
struct OctLaunchState {
  int wide_max_problems = 0;  // UVO_TUNE_OCT_WIDE_MAX (extractor.cpp sets the default)
};
bool octree_gauss_applies(const OctLaunchState& st, const Geom& g, int batch);
void launch_octree_gauss(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint8_t* d_pyr, uint8_t* d_blur, int64_t pyr_block, int4 taps, int sse2_rounding,
                         const uint32_t* d_cand_lo, int32_t* d_cursor, int32_t* d_fcount, int32_t* d_n_cell_list, uint8_t* d_cell_hi, uint32_t* d_cand_xy,
                         uint32_t* d_cand_sc, int64_t cand_block, int32_t* d_cand_count, uint32_t* d_pstate, uint32_t* d_sel_xy, uint32_t* d_sel_sc,
                         int32_t* d_sel_count, int batch, Level0View l0, const uint16_t* d_oct_tab);
int prepare_octree(const Geom& g);  // the part of the quad-tree launch that can fail (called before a batch's first kernel)
// the path tables of the closed-form quad-tree (octree_pyramid.hpp: path_xbits / path_ybits) are a constant of a level's geometry: built on the
// host once per geometry, g.lv[l].oct_tab_off entries into dst (uint16 each); the kernels copy a level's table into LDS instead of computing
// it per (frame, level) problem
void octree_fill_path_tables(const Geom& g, uint16_t* dst);
int launch_octree(hipStream_t s, OctLaunchState& st, const LevelGeom* d_lv, const Geom& g, const uint32_t* d_cand_lo, int32_t* d_cursor,
                   int32_t* d_fcount, int32_t* d_n_cell_list, uint8_t* d_cell_hi, uint32_t* d_cand_xy, uint32_t* d_cand_sc, int64_t cand_block, int32_t* d_cand_count, uint32_t* d_pstate,
                   uint32_t* d_sel_xy, uint32_t* d_sel_sc, int32_t* d_sel_count, int batch, const uint16_t* d_oct_tab);
// FAST mode feedback handed to k_assemble (its workgroup 0 sums the batch's fall-back cells and re-decides each level's mode)
struct FastAdapt {
  const int32_t* fcount;  // [batch][nlevels] fall-back cells per (frame, level), written by k_octree
  int32_t* tpass;         // [kMaxLevels] the lane's pass thresholds
  int32_t* last;          // [kMaxLevels] the batch's sums, kept for uvo_extractor_fast_state
  int adapt, fast_th;
};
void launch_assemble(hipStream_t s, const LevelGeom* d_lv, const Geom& g, FastAdapt fa, const uint32_t* d_sel_xy, const uint32_t* d_sel_sc,
                     const int32_t* d_sel_count,
                     const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int in_cap, int32_t* d_grid, int grid_rows, int grid_cols,
                     int min_px_dist, int full_detect, const int32_t* d_nfn, FinalSlot* d_flist, int32_t* d_n_final, int batch);
void launch_occupancy_grid(hipStream_t s, const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int in_cap, int min_px_dist, int grid_rows, int grid_cols,
                           int32_t* d_grid, int batch);
void launch_knn2(hipStream_t s, int pairs, int max_q, const uint8_t* d_q, const int32_t* d_nq, int nq_fixed, int q_stride, const uint8_t* d_t,
                 const int32_t* d_nt, int nt_fixed, int t_stride, const uint8_t* d_mask, int out_stride, int32_t* d_idx0, uint16_t* d_d0,
                 int32_t* d_idx1, uint16_t* d_d1);
void launch_medoid(hipStream_t s, const uint8_t* d_desc, const int32_t* d_offsets, int npoints, int32_t* d_idx, int32_t* d_med);
void launch_matrix(hipStream_t s, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, uint16_t* d_dist);
void launch_describe(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint8_t* d_pyr, const uint8_t* d_blur, int64_t pyr_block,
                     const FinalSlot* d_flist, const int32_t* d_n_final, const uvo_keypoint* d_in_kp, int in_cap, const float* d_pattern,
                     const uint32_t* d_patch, uvo_keypoint* d_out_kp, uint8_t* d_out_desc, int cap, int32_t* d_n_out, int batch, Level0View l0);

void launch_describe_direct(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint8_t* d_pyr, const uint8_t* d_blur, int64_t pyr_block, const uint32_t* d_sel_xy,
                            const uint32_t* d_sel_sc, const int32_t* d_sel_count, FastAdapt fa, const float* d_pattern, const uint32_t* d_patch,
                            uvo_keypoint* d_out_kp, uint8_t* d_out_desc, int cap, int32_t* d_n_out, int batch, Level0View l0);

void launch_clahe(hipStream_t s, const uint8_t* d_src, int w, int h, int64_t stride, int64_t frame_stride, int batch, int tiles_x, int tiles_y,
                  int tile_w, int tile_h, int clip_limit, float lut_scale, uint8_t* d_lut, uint8_t* d_dst, int64_t dst_stride,
                  int64_t dst_frame_stride);

}  // namespace uvo
