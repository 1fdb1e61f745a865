// Host side of the extractor: constructor tables, per-resolution geometry, device scratch, the launch sequence and
// the C-ABI entry points of include/uvo/uvo.h that replace USLAM::ORBextractor (src/ORBextractor.cc).
// No CPU fallback: every entry point needs a usable HIP device.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "fast_geom.hpp"
#include "profiler.hpp"
#include "tune_internal.h"
#include "uvo_math.hpp"

namespace uvo {

static thread_local char g_err[512] = "";
const char* hip_err_set(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return g_err;
}
int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

static const int8_t kPattern[1024] = {
#include "rbrief_pattern.inc"
};

static inline int cv_round_host(float v) { return (int)lrintf(v); }
static inline int cv_floor_host(float v) {
  int i = (int)v;
  return i - (i > v);
}

}  // namespace uvo

using namespace uvo;

// Per-batch scratch + stream.  With pipeline depth 2 consecutive uvo_extract_batch_device calls alternate between two
// lanes, so the latency-bound stages of one batch (quad-tree, sparse NMS, small pyramid levels) overlap with the
// throughput stages of the next.
constexpr int kMaxLanes = 4;
constexpr int kTilePyramidFrames = 8;  // batches up to this size build the pyramid in one k_pyr_tiles launch
constexpr int kFewFrames = 2;  // batches up to this size are the latency path (the FAST kernels cut their segments short for them: fast_rows_per_seg)

struct Lane {
  hipStream_t stream = nullptr;  // everything of a batch runs in this ONE in-order stream (no side streams: an error return leaves nothing to join)
  uint8_t *d_pyr = nullptr, *d_blur = nullptr;
  uint32_t *d_cand_xy = nullptr, *d_cand_sc = nullptr, *d_cand_lo = nullptr, *d_pstate = nullptr, *d_sel_xy = nullptr, *d_sel_sc = nullptr;
  int32_t *d_cand_count = nullptr, *d_sel_count = nullptr, *d_n_final = nullptr, *d_cor_n = nullptr, *d_cursor = nullptr;
  uint32_t* d_cor = nullptr;     // FAST corner lists, one region per k_fast_score wavefront
  uint8_t* d_cell_hi = nullptr;  // per cell: owns an NMS survivor >= fastTh
  // the lane's adaptive FAST mode: d_tpass[level] = threshold of the level's streaming pass (fastTh: threshold-adaptive two-pass form;
  // min(fastTh, 7): one pass + vote); d_fstat = the fall-back-cell sums k_octree turns into next batch's d_tpass (octree.hip)
  int32_t *d_tpass = nullptr, *d_fstat = nullptr, *d_fcount = nullptr;
  uint2* d_cell_list = nullptr;  // (cell, frame) of the fall-back cells of the batch in flight (k_fast_cells_list -> k_fast_cells)
  FinalSlot* d_flist = nullptr;
  Profiler prof;
  // staging of the asynchronous host-buffer form (allocated on first use): a lane's frames and results must not be touched by the
  // other lane's batch
  uint8_t* a_imgs = nullptr;
  uvo_keypoint* a_kp = nullptr;
  uint8_t* a_desc = nullptr;
  int32_t* a_n = nullptr;
  int a_batch = 0;  // frames of the batch in flight on this lane (0 = none)
  hipEvent_t a_uploaded = nullptr;  // recorded behind the lane's frame upload (asynchronous host form)
  // Device-resident batches never make the host wait, and a caller that enqueues them in a loop runs ahead of the device until the
  // runtime's own back-pressure stops it -- in bursts: the queue drains completely before the host is let go (0.9 - 1.5 ms with nothing in
  // flight every seven batches of 256 frames, tools/step_trace_summary.py).  The lane bounds its own depth instead: a batch is enqueued
  // only when the lane's last but one has finished, so at most two batches of a lane are ever outstanding.
  hipEvent_t done[2] = {nullptr, nullptr};
  unsigned n_enqueued = 0;
  // level 0 of the lane's last batch was read in place (no padded plane was written): what uvo_extractor_read_plane needs to make one
  const uint8_t* l0_src = nullptr;
  int64_t l0_stride = 0, l0_frame_stride = 0;
  int ring_used = 0;  // border pixels the lane's last batch wrote around levels >= 1 (what uvo_extractor_read_plane has to complete)
};

struct uvo_matcher;
extern "C" void uvo_matcher_follow_internal(uvo_matcher* m, hipStream_t s);
extern "C" void uvo_matcher_orphaned_internal(uvo_matcher* m);

struct uvo_extractor {
  uvo_extractor_cfg cfg;
  std::mutex followers_mu;               // attach / detach may come from the matcher's thread
  std::vector<uvo_matcher*> followers;  // matchers attached to this handle (uvo_matcher_attach_extractor): they enqueue in the current lane's stream
  int device = 0;
  Lane lane[kMaxLanes];
  int nlanes = 1, cur = 0;  // cur = the lane of the most recent batch
  OctLaunchState oct;       // quad-tree launch shape + what has been configured on this handle's device
  uint8_t* d_grid_score = nullptr;  // score plane of the Grider_FAST mode (allocated on first use)
  // constructor tables (src/ORBextractor.cc:463-511)
  std::vector<float> scale, inv_scale;
  std::vector<int> quota;
  int umax[16];
  int gtaps[4];
  // current geometry
  Geom geom;
  bool have_geom = false;
  std::vector<CellDesc> cells;
  std::vector<int32_t> cell_flag;  // per entry of a frame's cell-flag array (the full nRows x nCols grids of all levels): cell | level << 24, -1 = no cell
  int fast_mode = UVO_FAST_MODE_ADAPTIVE;
  int blur_rounding = UVO_BLUR_ROUNDING_SSE2;  // UVO_TUNE_BLUR_ROUNDING: what an x86-64 OpenCV 3.4 build (the reference's platform) executes
  // capacities fixed at create time (from max_width x max_height)
  int64_t cap_pyr_block = 0, cap_cand_block = 0;
  int cap_cells = 0, cap_sel_block = 0, cap_flist = 0, cap_xtab = 0, cap_ytab = 0;
  size_t cap_cor = 0, cap_cor_n = 0, cap_flags = 0;
  int last_batch = 0;
  // shared read-only tables
  LevelGeom* d_lv = nullptr;
  uint16_t* d_oct_tab = nullptr;  // the quad-tree's path tables of the current geometry (octree_fill_path_tables), per level at LevelGeom::oct_tab_off
  int cap_oct_tab = 0;
  CellDesc* d_cells = nullptr;
  int32_t* d_cell_flag = nullptr;
  ResizeCol* d_ctab = nullptr;
  uint8_t* d_clahe_lut = nullptr;  // [max_batch][tiles][256], grown on demand
  size_t clahe_lut_bytes = 0;
  uint8_t* d_clahe_out = nullptr;  // result of the host entry point (tight rows); stays valid for img == NULL calls
  int clahe_w = 0, clahe_h = 0;
  int resize_fast[kMaxLevels] = {0};  // per level: the 12-byte-window path of k_resize_level applies
  ResizeRow* d_rtab = nullptr;
  // the fused pyramid launches (pyramid.hip: k_pyr_tiles; plans: pyr_tiles.hpp).  pyr_form: UVO_TUNE_PYR_FORM.  One set of level groups per
  // geometry -- the latency shape (few frames: one launch of many small tiles), or a forced set (UVO_TUNE_PYR_TILE_GROUP); an empty set = the
  // per-level launches.  (Batches that fill the chip take the per-level launches: shallow groups of large tiles -- 4 % redundant pixels --
  // measured 0.26 - 0.31 ms against the launches' 0.16 at 256 frames, profiles/r05_pyr_tiles_ab.txt.)
  struct TileGroup {
    int first = 0, last = 0, tx = 0, ty = 0, threads = 256, rows = 4;  // rows: output rows per work item (1: only with 1024 threads)
    uint32_t lds = 0;
    PyrTileLevel* d_plan = nullptr;
  };
  std::vector<TileGroup> tile_groups;
  uint32_t pyr_tiles_max_lds = 64 * 1024;  // LDS a tile plan may use on this device (prepare_pyr_tiles); plans that need more fall back to the per-level launches
  std::vector<uint32_t> tile_spec;  // forced groups: first << 16 | tx << 8 | ty | (1024 threads) << 24 | (1024 threads, single-row items) << 25, ascending first levels
  int pyr_form = UVO_PYR_FORM_AUTO;
  std::vector<ResizeCol> ctab_host;  // the resize tables of the current geometry (the plans are compiled from them)
  std::vector<ResizeRow> rtab_host;
  int pyr_ring = 4;          // UVO_TUNE_PYR_RING: the chain's resize launches write the ROI + this many pixels around it (0: the whole 16-pixel pad)
  int level0_inplace = 1;    // UVO_TUNE_LEVEL0_INPLACE: read level 0 from the caller's image instead of copying it into a padded plane (when it can be)
  int zero_copy_out = 1;     // UVO_TUNE_ZERO_COPY_OUT: host-buffer calls of up to 16 frames have k_describe write into page-locked host memory
  int spin_wait = 1;         // UVO_TUNE_SPIN_WAIT: those calls, and uvo_extractor_synchronize behind a small batch, poll the stream instead of sleeping
  int few_frames_shape = 1;  // UVO_TUNE_FEW_FRAMES: FullDetect batches of up to kFewFrames frames take the short launch chain (no k_assemble: k_describe finds its slots itself)
  int fuse_blur_tree = 1;    // UVO_TUNE_FUSE_BLUR_TREE: quad-tree and blur as one launch when the batch takes the 256-thread quad-tree form
  float* d_pattern = nullptr;   // 256 point pairs (x0, y0, x1, y1) of the rBRIEF pattern as floats
  uint32_t* d_patch = nullptr;  // 256 byte masks: which of the 4 pixels of an orientation-patch dword lie inside the circle
  // staging for the host-buffer entry points
  uint8_t* d_imgs = nullptr;
  uvo_keypoint *d_out_kp = nullptr, *d_in_kp = nullptr;
  uint8_t* d_out_desc = nullptr;
  int32_t *d_n_out = nullptr, *d_n_in = nullptr, *d_nfn = nullptr, *d_grid = nullptr;
  size_t grid_bytes = 0;
  // pinned host staging of the host-buffer entry points' small inputs / outputs (copies to and from pageable caller
  // memory stall the stream once per call; a pinned bounce keeps them asynchronous behind one wait)
  uint8_t* h_pin = nullptr;
  uint8_t* h_pin_dev = nullptr;  // the same memory as the device addresses it (k_describe writes small batches' results straight into it)
  size_t pin_bytes = 0;
  // asynchronous host form: the event behind the most recent frame upload of ANY lane.  The next upload waits for it, so that uploads
  // follow one another instead of sharing the link: two lanes that upload at the same time finish together, then compute together,
  // then download together -- link idle while the GPU works and the GPU idle while the link works (measured: 2.36 instead of 1.87 ms
  // per 256-frame job, and the in-phase pattern is stable once entered).  One after the other, lane B's upload runs under lane A's
  // kernels whatever the kernels' durations are.
  hipEvent_t last_upload = nullptr;
};

namespace uvo {

static int sync_all_lanes(uvo_extractor* h);
static int alloc_lane(uvo_extractor* h, int li);

// Waits for a stream.  spin: the latency path -- hipStreamSynchronize gives up its busy wait after a few microseconds and sleeps on an
// interrupt, whose wake-up costs more than a whole stage of a single frame's chain; polling the stream's state keeps the host on the
// spot for the ~100 us a frame takes (bounded: after kSpinWaitUs the blocking wait takes over).
constexpr int kSpinWaitUs = 400;
static hipError_t wait_stream(hipStream_t s, bool spin) {
  if (spin) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e != hipErrorNotReady) return e;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kSpinWaitUs) break;
    }
  }
  return hipStreamSynchronize(s);
}

// (Re)starts lane li's per-level FAST mode.  Adaptive and two-pass start threshold-adaptive (stream at fastTh, sparse literal-7 pass);
// single-pass streams at min(fastTh, 7) and votes.  With fastTh <= 7 the second call of src/ORBextractor.cc:797 can find nothing the
// first did not, so there is only one form.
static int set_lane_fast_mode(uvo_extractor* h, int li) {
  int32_t t[kMaxLevels];
  const int th = h->cfg.fast_th;
  for (int l = 0; l < kMaxLevels; ++l) t[l] = (th > 7 && h->fast_mode != UVO_FAST_MODE_SINGLE_PASS) ? th : (th < 7 ? th : 7);
  UVO_HIP_CHECK(hipMemcpy(h->lane[li].d_tpass, t, sizeof(t), hipMemcpyHostToDevice));
  return UVO_OK;
}

// ORBextractor::ORBextractor: src/ORBextractor.cc:458-512
static void build_ctor_tables(uvo_extractor* h) {
  const int nl = h->cfg.nlevels;
  const double scaleFactor = (double)h->cfg.scale_factor;  // member is double (include/ORBextractor.h:79)
  h->scale.assign(nl, 1.f);
  h->inv_scale.assign(nl, 1.f);
  for (int i = 1; i < nl; ++i) h->scale[i] = (float)(h->scale[i - 1] * scaleFactor);
  const float invScaleFactor = (float)(1.0f / scaleFactor);
  for (int i = 1; i < nl; ++i) h->inv_scale[i] = h->inv_scale[i - 1] * invScaleFactor;
  h->quota.assign(nl, 0);
  const float factor = (float)(1.0 / scaleFactor);
  float nDesired = h->cfg.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; ++l) {
    h->quota[l] = cv_round_host(nDesired);
    sum += h->quota[l];
    nDesired *= factor;
  }
  h->quota[nl - 1] = std::max(h->cfg.nfeatures - sum, 0);
  // umax (:494-511), HALF_PATCH_SIZE = 15
  const int HP = 15;
  int v, v0, vmax = cv_floor_host(HP * sqrtf(2.f) / 2 + 1);
  int vmin = (int)ceilf(HP * sqrtf(2.f) / 2);
  const double hp2 = HP * HP;
  for (v = 0; v <= vmax; ++v) h->umax[v] = (int)lrint(sqrt(hp2 - v * v));
  for (v = HP, v0 = 0; v >= vmin; --v) {
    while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
    h->umax[v] = v0;
    ++v0;
  }
  // Gaussian taps: cv::getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256)
  float cf[7];
  double s = 0;
  for (int i = 0; i < 7; ++i) {
    double x = i - 3.0;
    cf[i] = (float)std::exp(-0.5 / 4.0 * x * x);
    s += cf[i];
  }
  s = 1. / s;
  for (int i = 0; i < 4; ++i) h->gtaps[i] = cv_round_host((float)(cf[i] * s) * 256.f);
}

// Geometry of one resolution: pyramid sizes (:966-969), detection window and FAST cells (:755-790),
// quad-tree roots (:1010-1012), scratch offsets.
static int build_geom(const uvo_extractor* h, int width, int height, Geom& g, std::vector<CellDesc>& cells, std::vector<int32_t>* cell_flag = nullptr) {
  const int nl = h->cfg.nlevels;
  memset(&g, 0, sizeof(g));
  g.width = width, g.height = height, g.nlevels = nl;
  cells.clear();
  if (cell_flag) cell_flag->clear();
  int64_t off = 0, coff = 0;
  int soff = 0, xt = 0, yt = 0, ot = 0, flag_base = 0;
  for (int l = 0; l < nl; ++l) {
    LevelGeom& L = g.lv[l];
    L.w = cv_round_host((float)width * h->inv_scale[l]);
    L.h = cv_round_host((float)height * h->inv_scale[l]);
    if (L.w < 56 || L.h < 56 || L.w > 4096 || L.h > 4096) return fail(UVO_E_UNSUPPORTED, "pyramid level outside 56..4096 px");
    L.pw = L.w + 2 * kPad, L.ph = L.h + 2 * kPad;
    L.pitch = (L.pw + kPyrPitchAlign - 1) / kPyrPitchAlign * kPyrPitchAlign;  // whole cache lines: a line never holds bytes of two rows (pyr_schedule.hpp)
    L.plane_off = off;
    off += (int64_t)L.pitch * ((L.ph + 7) & ~7);  // whole 16 x 8 tiles: the blurred plane is stored tiled (gauss.hip), same offsets for both
    off = (off + 255) / 256 * 256;
    L.bw = L.w - 2 * kMinBorder, L.bh = L.h - 2 * kMinBorder;
    const float fw = (float)L.bw, fh = (float)L.bh;
    L.nCols = (int)(fw / 30.f), L.nRows = (int)(fh / 30.f);
    L.wCell = (int)ceilf(fw / L.nCols), L.hCell = (int)ceilf(fh / L.nRows);
    L.cell_base = (int)cells.size();
    const int maxBX = L.w - kMinBorder, maxBY = L.h - kMinBorder;
    int cap = 0;
    for (int i = 0; i < L.nRows; ++i) {
      const int iniY = kMinBorder + i * L.hCell;
      int maxY = iniY + L.hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = maxBY;
      for (int j = 0; j < L.nCols; ++j) {
        const int iniX = kMinBorder + j * L.wCell;
        int maxX = iniX + L.wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = maxBX;
        CellDesc c;
        c.level = (int16_t)l;
        c.x0 = (int16_t)iniX, c.y0 = (int16_t)iniY;
        c.rw = (int16_t)(maxX - iniX), c.rh = (int16_t)(maxY - iniY);
        c.ox = (int16_t)(j * L.wCell), c.oy = (int16_t)(i * L.hCell);
        c.pad = 0;
        if (c.rw > 66 || c.rh > 66) return fail(UVO_E_UNSUPPORTED, "FAST cell larger than 66 px");
        const int iw = c.rw - 6, ih = c.rh - 6;
        if (iw <= 0 || ih <= 0) continue;  // FAST on an ROI without interior finds nothing
        cap += ((iw + 1) / 2) * ((ih + 1) / 2);
        cells.push_back(c);
        if (cell_flag) {
          cell_flag->resize((size_t)flag_base + (size_t)L.nRows * L.nCols, -1);
          (*cell_flag)[flag_base + i * L.nCols + j] = (int32_t)(cells.size() - 1) | (l << 24);
        }
      }
    }
    flag_base += L.nRows * L.nCols;  // the same running base as fast_levels() (fast.hip)
    if (cell_flag) cell_flag->resize((size_t)flag_base, -1);
    if (cells.size() >= (1u << 24)) return fail(UVO_E_UNSUPPORTED, "more than 2^24 FAST cells per frame");
    L.n_cells = (int)cells.size() - L.cell_base;
    L.quota = h->quota[l];
    if (L.quota > kMaxOctN) return fail(UVO_E_UNSUPPORTED, "per-level feature quota above the quad-tree kernel's capacity");
    L.cand_cap = cap;
    L.cand_off = coff;
    coff += (cap + 63) / 64 * 64;
    L.nIni = (int)roundf((float)L.bw / (float)L.bh);
    if (L.nIni < 1 || L.nIni > 64) return fail(UVO_E_UNSUPPORTED, "image aspect ratio outside the quad-tree's range");
    // DistributeOctTree returns at most quota + 3 nodes once it is in its careful phase, but the first pass splits all nIni
    // roots unconditionally: up to 4 * nIni nodes whatever the quota (wide images with few features)
    L.sel_cap = std::max(L.quota, 4 * L.nIni) + 4;
    L.sel_off = soff;
    soff += L.sel_cap;
    L.hX = (float)L.bw / (float)L.nIni;
    L.scale = h->scale[l];
    L.patch_size = (float)(int)(31 * h->scale[l]);
    L.xtab_off = xt, L.ytab_off = yt;
    if (l > 0) xt += L.pitch, yt += (L.ph + 3) & ~3;  // row tables are padded to whole groups of 4 rows (k_resize_level reads a group at once)
    L.oct_tab_off = ot, L.pad_ = 0;
    ot += (L.bw + L.bh + 1) & ~1;  // (an even number of 2-byte entries: the kernels copy a table as dwords)
  }
  g.total_cells = (int)cells.size();
  g.pyr_block = off;
  g.cand_block = coff;
  g.sel_block = soff;
  g.flist_cap = soff + h->cfg.max_input_keypoints;
  return UVO_OK;
}

// cv::resize coefficient tables of every level >= 1 (pyr_tiles.hpp: pyr_build_level_tables), concatenated at the offsets build_geom assigned
static void build_resize_tables(const Geom& g, std::vector<ResizeCol>& ctab, std::vector<ResizeRow>& rtab, int* fast_ok) {
  ctab.clear(), rtab.clear();
  for (int l = 1; l < g.nlevels; ++l) pyr_build_level_tables(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, g.lv[l].pitch, ctab, rtab, &fast_ok[l]);
}

template <class T>
static int dev_alloc(T** p, size_t n) {
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) {
    hip_err_set(e, "hipMalloc");
    return e == hipErrorOutOfMemory ? UVO_E_NOMEM : UVO_E_HIP;
  }
  return UVO_OK;
}

static void free_tile_groups(uvo_extractor* h) {
  for (auto& G : h->tile_groups)
    if (G.d_plan) (void)hipFree(G.d_plan);
  h->tile_groups.clear();
}

// Compiles the plans of one set of level groups ({first level, tx, ty, threads} each, ascending) for geometry g.  A set that cannot be
// built (a level outside the 12-byte tap window, a tile larger than the LDS) stays empty: the batch then takes the per-level launches.
static int build_tile_set(uvo_extractor* h, const Geom& g, const std::vector<uint32_t>& spec, std::vector<uvo_extractor::TileGroup>& out) {
  out.clear();
  if (spec.empty() || g.nlevels < 2) return UVO_OK;
  PyrTileDims dims[kMaxLevels];
  const ResizeCol* cp[kMaxLevels] = {nullptr};
  const ResizeRow* rp[kMaxLevels] = {nullptr};
  for (int l = 0; l < g.nlevels; ++l) {
    dims[l] = PyrTileDims{g.lv[l].w, g.lv[l].h, g.lv[l].pitch};
    if (l > 0) {
      if (!h->resize_fast[l]) return UVO_OK;
      cp[l] = h->ctab_host.data() + g.lv[l].xtab_off, rp[l] = h->rtab_host.data() + g.lv[l].ytab_off;
    }
  }
  std::vector<uvo_extractor::TileGroup> set;
  for (size_t k = 0; k < spec.size(); ++k) {
    uvo_extractor::TileGroup G;
    G.first = (int)((spec[k] >> 16) & 0xff), G.tx = (int)((spec[k] >> 8) & 0xff), G.ty = (int)(spec[k] & 0xff), G.threads = (spec[k] >> 24) & 3 ? 1024 : 256, G.rows = (spec[k] >> 25) & 1 ? 1 : 4;
    G.last = k + 1 < spec.size() ? (int)((spec[k + 1] >> 16) & 0xff) - 1 : g.nlevels - 1;
    if (G.first >= g.nlevels) break;  // (a spec written for more levels than this handle has)
    G.last = std::min(G.last, g.nlevels - 1);
    PyrTilePlan P;
    if (G.first < 1 || G.last < G.first || (k == 0 && G.first != 1) || !pyr_tile_plan_build(dims, g.nlevels, G.first, G.last, cp, rp, 4, G.tx, G.ty, h->pyr_tiles_max_lds, P)) {
      for (auto& X : set)
        if (X.d_plan) (void)hipFree(X.d_plan);
      return UVO_OK;
    }
    G.lds = P.lds_bytes;
    int rc = dev_alloc(&G.d_plan, P.lv.size());
    if (rc == UVO_OK && hipMemcpy(G.d_plan, P.lv.data(), P.lv.size() * sizeof(PyrTileLevel), hipMemcpyHostToDevice) != hipSuccess) rc = fail(UVO_E_HIP, "plan upload failed");
    if (rc != UVO_OK) {
      if (G.d_plan) (void)hipFree(G.d_plan);
      for (auto& X : set)
        if (X.d_plan) (void)hipFree(X.d_plan);
      return rc;
    }
    set.push_back(G);
  }
  out.swap(set);
  return UVO_OK;
}

// number of tiles along an axis of `len` pixels for tiles of about `target` pixels
static inline uint32_t tiles_for(int len, int target) { return (uint32_t)std::min(255, std::max(1, (len + target / 2) / target)); }

// The default set of a geometry: the latency shape (a handful of frames cannot fill the chip: as many workgroups as CUs, ONE launch of
// 1024-thread workgroups with single-row work items -- the halo of a deep group is paid in redundant pixels, which idle CUs have to spare:
// measured against two and three launches and against 256-thread workgroups, profiles/r05_latency_ab.txt).  Called with every lane idle.
static int build_tile_groups(uvo_extractor* h, const Geom& g) {
  free_tile_groups(h);
  if (!h->tile_spec.empty()) return build_tile_set(h, g, h->tile_spec, h->tile_groups);
  if (g.nlevels < 2) return UVO_OK;
  const std::vector<uint32_t> lat{1u << 25 | 1u << 16 | tiles_for(g.lv[1].w, 34) << 8 | tiles_for(g.lv[1].h, 27)};  // 16 x 16 tiles at 640 x 512
  return build_tile_set(h, g, lat, h->tile_groups);
}

static int set_geometry(uvo_extractor* h, int width, int height) {
  if (h->have_geom && h->geom.width == width && h->geom.height == height) return UVO_OK;
  Geom g;
  std::vector<CellDesc> cells;
  std::vector<int32_t> cell_flag;
  int rc = build_geom(h, width, height, g, cells, &cell_flag);
  if (rc) return rc;
  if (g.pyr_block > h->cap_pyr_block || g.cand_block > h->cap_cand_block || g.total_cells > h->cap_cells || g.sel_block > h->cap_sel_block ||
      g.flist_cap > h->cap_flist)
    return fail(UVO_E_BADARG, "image larger than the handle was sized for");
  for (int b = 1; b <= h->cfg.max_batch; b = b < 16 ? b + 1 : h->cfg.max_batch) {
    const int rps = fast_rows_per_seg(b);
    const size_t it = (size_t)fast_items_per_frame(g, rps);
    if ((size_t)b * it * (size_t)FS_REGION_ENTRIES > h->cap_cor || (size_t)b * it > h->cap_cor_n ||
        (size_t)h->cfg.max_batch * fast_flags_per_frame(g) > h->cap_flags)
      return fail(UVO_E_BADARG, "image larger than the handle was sized for");
    if (b == h->cfg.max_batch) break;
  }
  std::vector<ResizeCol> ctab;
  std::vector<ResizeRow> rtab;
  build_resize_tables(g, ctab, rtab, h->resize_fast);
  if ((int)ctab.size() > h->cap_xtab || (int)rtab.size() > h->cap_ytab) return fail(UVO_E_BADARG, "image larger than the handle was sized for");
  // in-flight work may still read the old tables
  {
    int rcs = sync_all_lanes(h);
    if (rcs) return rcs;
  }
  {
    const LevelGeom& last = g.lv[g.nlevels - 1];
    const int n_oct = last.oct_tab_off + ((last.bw + last.bh + 1) & ~1);
    if (n_oct > h->cap_oct_tab) return fail(UVO_E_BADARG, "image larger than the handle was sized for");
    std::vector<uint16_t> oct_tab((size_t)n_oct, 0);
    octree_fill_path_tables(g, oct_tab.data());
    UVO_HIP_CHECK(hipMemcpy(h->d_oct_tab, oct_tab.data(), oct_tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  UVO_HIP_CHECK(hipMemcpy(h->d_lv, g.lv, sizeof(LevelGeom) * g.nlevels, hipMemcpyHostToDevice));
  UVO_HIP_CHECK(hipMemcpy(h->d_cells, cells.data(), sizeof(CellDesc) * cells.size(), hipMemcpyHostToDevice));
  UVO_HIP_CHECK(hipMemcpy(h->d_cell_flag, cell_flag.data(), sizeof(int32_t) * cell_flag.size(), hipMemcpyHostToDevice));
  if (!ctab.empty()) {
    UVO_HIP_CHECK(hipMemcpy(h->d_ctab, ctab.data(), ctab.size() * sizeof(ResizeCol), hipMemcpyHostToDevice));
    UVO_HIP_CHECK(hipMemcpy(h->d_rtab, rtab.data(), rtab.size() * sizeof(ResizeRow), hipMemcpyHostToDevice));
  }
  h->ctab_host.swap(ctab), h->rtab_host.swap(rtab);
  {
    int rct = build_tile_groups(h, g);
    if (rct) return rct;
  }
  h->geom = g;
  h->cells = cells;
  h->cell_flag = cell_flag;
  h->have_geom = true;
  return UVO_OK;
}

struct ProfScope : Profiler::Scope {
  ProfScope(uvo_extractor* h, const char* name) : Profiler::Scope(&h->lane[h->cur].prof, name, h->lane[h->cur].stream) {}
};

// The lane the next batch runs on.  With pipeline depth > 1 consecutive batches alternate; whoever stages inputs for a batch
// (uploads, CLAHE) must enqueue them on this lane's stream, and read results back from it.
static inline int next_lane(const uvo_extractor* h) { return h->nlanes > 1 ? (h->cur + 1) % h->nlanes : h->cur; }

// The launch sequence of one batch: everything on lane `li`'s stream, nothing synchronous.  Makes `li` the current lane.
static int run_batch_device(uvo_extractor* h, int li, int batch, const uint8_t* d_imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                            const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int32_t* d_grid2d, int grid_rows, int grid_cols,
                            int min_px_dist, int full_detect, const int32_t* d_nfn, uvo_keypoint* d_out_kp, uint8_t* d_out_desc, int cap,
                            int32_t* d_n_out) {
  if (!h || !d_imgs || !d_out_kp || !d_out_desc || !d_n_out) return fail(UVO_E_BADARG, "null pointer");
  if (batch < 1 || batch > h->cfg.max_batch) return fail(UVO_E_BADARG, "batch outside 1..max_batch");
  if (width < 1 || height < 1 || stride < width || cap < 1) return fail(UVO_E_BADARG, "bad image size / stride / cap");
  if (!full_detect && (!d_grid2d || !d_nfn || min_px_dist < 1 || grid_rows < 1 || grid_cols < 1))
    return fail(UVO_E_BADARG, "top-up mode needs grid2d, num_feats_needed and min_px_dist >= 1");
  // the occupancy filter indexes grid_2d(int(y / d), int(x / d)) for every pixel position (src/ORBextractor.cc:884-891; the call
  // site allocates rows / d + 2 by cols / d + 2, src/Tracking.cc:930-934): a smaller grid would be indexed out of bounds
  if (!full_detect && (grid_rows <= (height - 1) / min_px_dist || grid_cols <= (width - 1) / min_px_dist))
    return fail(UVO_E_BADARG, "grid2d smaller than ceil(image / min_px_dist)");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  int rc = set_geometry(h, width, height);
  if (rc) return rc;
  if ((rc = prepare_octree(h->geom)) != UVO_OK) return rc;  // nothing below this line fails once the first kernel is in the stream
  h->cur = li;
  Lane& L = h->lane[li];
  hipStream_t s = L.stream;
  {  // attached matchers work behind THIS batch (and only this one: the ordering contract of uvo_matcher_attach_extractor)
    std::lock_guard<std::mutex> lk(h->followers_mu);
    for (uvo_matcher* m : h->followers) uvo_matcher_follow_internal(m, s);
  }
  h->last_batch = batch;
  const Geom& g = h->geom;
  hipEvent_t& done = L.done[L.n_enqueued & 1];  // recorded behind the lane's last but one batch
  if (done) UVO_HIP_CHECK(hipEventSynchronize(done));
  else UVO_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  // Level 0 in place: every reader of level 0 (the FAST kernels, the blur, the orientation patch, the resize to level 1) takes the caller's
  // image itself -- the blur reflects its 3-pixel border on the fly -- and cv::copyMakeBorder of :996 (182 MB per 257 frames at 640 x 512)
  // is never materialised.  Needs dword-aligned rows of a width that is a multiple of 4 (a lane's four pixels are then wholly inside the
  // image or wholly border), the launch chain, and no caller keypoints (their orientation patch may reach into the border).  The image
  // must stay unchanged until the batch is complete (it always had to stay valid that long).
  const bool inplace = h->level0_inplace && width % 4 == 0 && stride % 4 == 0 && frame_stride % 4 == 0 &&
                       (uintptr_t)d_imgs % 4 == 0 && !(d_in_kp && d_n_in) && width >= 64 && height >= 64 &&
                       // the kernels address in-place rows as __umul24(row, pitch) + a 32-bit lane offset: a wider stride (an ROI of a large
                       // mosaic) takes the padded copy instead
                       stride <= (1 << 20) && (int64_t)(height + 2 * kPad) * stride < (int64_t)1 << 31;
  Level0View l0{nullptr, 0, 0, 0};
  if (inplace) l0 = Level0View{d_imgs - (int64_t)kPad * stride - kPad, (int64_t)frame_stride, (int)stride, 0};
  L.l0_src = inplace ? d_imgs : nullptr, L.l0_stride = stride, L.l0_frame_stride = frame_stride;
  L.ring_used = h->pyr_ring;
  const Level0View no_l0{nullptr, 0, 0, 0};
  {
    // ComputePyramid (src/ORBextractor.cc:963-1004): a launch per group of levels (k_pyr_tiles), or one per level
    if (!inplace) {
      ProfScope p(h, "k_pad_level0");
      launch_pad_level0(s, d_imgs, width, height, stride, frame_stride, L.d_pyr, g.pyr_block, g.lv[0], batch);
    }
    const std::vector<uvo_extractor::TileGroup>* tiles = nullptr;
    if (h->pyr_ring == 4 && !h->tile_groups.empty() && (h->pyr_form == UVO_PYR_FORM_TILES || (h->pyr_form == UVO_PYR_FORM_AUTO && batch <= kTilePyramidFrames)))
      tiles = &h->tile_groups;
    if (tiles) {
      for (const auto& G : *tiles) {
        ProfScope p(h, "k_pyr_tiles");
        launch_pyr_tiles(s, L.d_pyr, g.pyr_block, G.d_plan, g, h->d_ctab, h->d_rtab, l0, G.first, G.last, G.tx * G.ty, G.lds, G.threads, G.rows, batch);
      }
    } else {
      for (int l = 1; l < g.nlevels; ++l) {
        ProfScope p(h, "k_resize_level");
        launch_resize_level(s, L.d_pyr, g.pyr_block, g.lv[l - 1], g.lv[l], h->d_ctab + g.lv[l].xtab_off, h->d_rtab + g.lv[l].ytab_off, h->resize_fast[l],
                            batch, l == 1 ? l0 : no_l0, h->pyr_ring);
      }
    }
  }
  const bool fused_tree = h->fuse_blur_tree && octree_gauss_applies(h->oct, g, batch);
  // A frame or two (the per-frame latency path, src/Tracking.cc:946): every stage is a chain of dependent phases on a nearly empty chip,
  // so the number of stages is what counts -- a FullDetect call has no k_assemble launch (k_describe finds its slots itself).  (One FAST
  // pass at 7 with the vote in the quad-tree instead of the sparse second launch: the pass takes 7.7 us longer, the launch it saves 8.)
  const bool few = batch <= kFewFrames && h->few_frames_shape;
  const int32_t* tpass = L.d_tpass;
  const int4 gtaps = make_int4(h->gtaps[0], h->gtaps[1], h->gtaps[2], h->gtaps[3]);
  {  // the per-cell threshold vote + candidate emit run inside k_octree
    ProfScope p(h, "k_fast_score");
    launch_fast_score(s, L.d_pyr, g.pyr_block, g, h->cfg.fast_th, tpass, L.d_cor, L.d_cell_hi, L.d_cand_xy, L.d_cand_sc, L.d_cand_lo, g.cand_block,
                      L.d_cursor, batch, l0);
  }
  if (h->cfg.fast_th > 7 && h->fast_mode != UVO_FAST_MODE_SINGLE_PASS) {
    // second call of src/ORBextractor.cc:797 for the cells of threshold-adaptive levels that the pass at fastTh left empty (nearly all
    // wavefronts find nothing to do on textured frames).  With the mode pinned to one pass no level can be adaptive: not launched.
    ProfScope p(h, "k_fast_cells");
    launch_fast_cells(s, L.d_pyr, g.pyr_block, g, h->d_cells, h->d_cell_flag, tpass, L.d_cell_hi, L.d_cell_list, L.d_fstat + kMaxLevels, L.d_cand_xy,
                      L.d_cand_sc, g.cand_block, L.d_cursor, batch, l0);
  }
  if (fused_tree) {
    // the quad-tree (a chain of dependent phases per (frame, level)) and the blur (a streaming kernel) read nothing of each other:
    // one grid, the quad-tree problems first, and the blur fills the issue slots they leave idle
    ProfScope p(h, "k_octree_gauss");
    launch_octree_gauss(s, h->d_lv, g, L.d_pyr, L.d_blur, g.pyr_block, gtaps, h->blur_rounding, L.d_cand_lo, L.d_cursor,
                        L.d_fcount, L.d_fstat + kMaxLevels, L.d_cell_hi, L.d_cand_xy, L.d_cand_sc, g.cand_block, L.d_cand_count, L.d_pstate, L.d_sel_xy, L.d_sel_sc, L.d_sel_count,
                        batch, l0, h->d_oct_tab);
  } else {
    {
      ProfScope p(h, "k_gauss7");
      launch_gauss7(s, L.d_pyr, L.d_blur, g.pyr_block, h->d_lv, g, gtaps, batch, h->blur_rounding, l0);
    }
    {
      ProfScope p(h, "k_octree");
      rc = launch_octree(s, h->oct, h->d_lv, g, L.d_cand_lo, L.d_cursor, L.d_fcount, L.d_fstat + kMaxLevels, L.d_cell_hi, L.d_cand_xy, L.d_cand_sc, g.cand_block, L.d_cand_count,
                         L.d_pstate, L.d_sel_xy, L.d_sel_sc, L.d_sel_count, batch, h->d_oct_tab);
      if (rc) return rc;
    }
  }
  const bool direct = few && full_detect && !(d_in_kp && d_n_in);
  const FastAdapt fa{L.d_fcount, L.d_tpass, L.d_fstat, h->fast_mode == UVO_FAST_MODE_ADAPTIVE ? 1 : 0, h->cfg.fast_th};
  if (!direct) {
    ProfScope p(h, "k_assemble");
    launch_assemble(s, h->d_lv, g, fa, L.d_sel_xy, L.d_sel_sc, L.d_sel_count, d_in_kp, d_n_in, h->cfg.max_input_keypoints, d_grid2d, grid_rows, grid_cols,
                    min_px_dist, full_detect, d_nfn, L.d_flist, L.d_n_final, batch);
  }
  {
    ProfScope p(h, "k_describe");
    if (direct)
      launch_describe_direct(s, h->d_lv, g, L.d_pyr, L.d_blur, g.pyr_block, L.d_sel_xy, L.d_sel_sc, L.d_sel_count, fa, h->d_pattern, h->d_patch, d_out_kp,
                             d_out_desc, cap, d_n_out, batch, l0);
    else
      launch_describe(s, h->d_lv, g, L.d_pyr, L.d_blur, g.pyr_block, L.d_flist, L.d_n_final, d_in_kp, h->cfg.max_input_keypoints, h->d_pattern,
                      h->d_patch, d_out_kp, d_out_desc, cap, d_n_out, batch, l0);
  }
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipEventRecord(done, s));
  L.n_enqueued++;
  return UVO_OK;
}

}  // namespace uvo

namespace uvo {
static int alloc_lane(uvo_extractor* h, int li) {
  Lane& L = h->lane[li];
  if (L.stream) return UVO_OK;
  const size_t B = (size_t)h->cfg.max_batch;
  hipError_t e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    hip_err_set(e, "hipStreamCreate");
    return UVO_E_HIP;
  }
  int rc;
#define AL(call) \
  if ((rc = (call)) != UVO_OK) return rc;
  AL(dev_alloc(&L.d_pyr, B * h->cap_pyr_block));
  AL(dev_alloc(&L.d_blur, B * h->cap_pyr_block + 256));  // + slack: k_describe reads whole dwords up to 3 bytes past a row end
  AL(dev_alloc(&L.d_cand_xy, B * h->cap_cand_block));
  AL(dev_alloc(&L.d_cand_sc, B * h->cap_cand_block));
  AL(dev_alloc(&L.d_cand_lo, B * h->cap_cand_block));
  AL(dev_alloc(&L.d_cursor, B * kMaxLevels * 2));
  AL(dev_alloc(&L.d_pstate, B * h->cap_cand_block));
  AL(dev_alloc(&L.d_sel_xy, B * h->cap_sel_block));
  AL(dev_alloc(&L.d_sel_sc, B * h->cap_sel_block));
  AL(dev_alloc(&L.d_cand_count, B * kMaxLevels));
  AL(dev_alloc(&L.d_sel_count, B * kMaxLevels));
  AL(dev_alloc(&L.d_n_final, B));
  AL(dev_alloc(&L.d_flist, B * h->cap_flist));
  AL(dev_alloc(&L.d_cor, h->cap_cor));
  AL(dev_alloc(&L.d_cor_n, h->cap_cor_n));
  AL(dev_alloc(&L.d_cell_hi, h->cap_flags));
  AL(dev_alloc(&L.d_tpass, (size_t)kMaxLevels));
  // the last batch's fall-back cells per level, then the length of d_cell_list; per (frame, level) counts of the batch in flight
  AL(dev_alloc(&L.d_fstat, (size_t)kMaxLevels + 4));
  AL(dev_alloc(&L.d_fcount, B * kMaxLevels));
  AL(dev_alloc(&L.d_cell_list, B * (size_t)h->cap_cells));
  if ((rc = set_lane_fast_mode(h, li)) != UVO_OK) return rc;
  if (hipMemset(L.d_fstat, 0, (kMaxLevels + 4) * sizeof(int32_t)) != hipSuccess) return fail(UVO_E_HIP, "hipMemset failed");
  // the cell flags and the fill cursors are zero between calls: k_fast_score sets / advances them, k_octree clears what it has consumed
  if (hipMemset(L.d_cell_hi, 0, h->cap_flags) != hipSuccess || hipMemset(L.d_cursor, 0, B * kMaxLevels * 2 * sizeof(int32_t)) != hipSuccess)
    return fail(UVO_E_HIP, "hipMemset failed");
#undef AL
  return UVO_OK;
}
static int sync_all_lanes(uvo_extractor* h) {
  for (int i = 0; i < kMaxLanes; ++i)
    if (h->lane[i].stream) UVO_HIP_CHECK(hipStreamSynchronize(h->lane[i].stream));
  return UVO_OK;
}
}  // namespace uvo

extern "C" {

const char* uvo_last_error(void) { return uvo::g_err; }

int uvo_device_info(int device, char* dst, int cap) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(UVO_E_NODEVICE, "no HIP device");
  if (device < 0 || device >= n) return fail(UVO_E_BADARG, "device ordinal out of range");
  hipDeviceProp_t p;
  UVO_HIP_CHECK(hipGetDeviceProperties(&p, device));
  snprintf(dst, cap, "uvo 0.1 %s %s CUs=%d", p.gcnArchName, p.name, p.multiProcessorCount);
  return UVO_OK;
}

int uvo_extractor_create(const uvo_extractor_cfg* cfg, uvo_extractor** out) {
  if (!cfg || !out) return fail(UVO_E_BADARG, "null pointer");
  *out = nullptr;
  if (cfg->nfeatures < 1 || cfg->nlevels < 1 || cfg->nlevels > kMaxLevels || !(cfg->scale_factor > 1.0f) || cfg->fast_th < 0 ||
      cfg->max_width < 1 || cfg->max_height < 1 || cfg->max_batch < 1 || cfg->max_input_keypoints < 0)
    return fail(UVO_E_BADARG, "bad extractor configuration");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(UVO_E_NODEVICE, "no HIP device available (no CPU fallback exists)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(UVO_E_BADARG, "device ordinal out of range");
  uvo_extractor* h = new uvo_extractor();
  h->cfg = *cfg;
  h->cfg.fast_th = std::min(cfg->fast_th, 255);  // cv::FAST clamps its threshold to 0 .. 255 (no byte differs from another by more)
  h->device = cfg->device;
  build_ctor_tables(h);
  Geom g;
  std::vector<CellDesc> cells;
  int rc = build_geom(h, cfg->max_width, cfg->max_height, g, cells);
  if (rc) {
    delete h;
    return rc;
  }
  // capacities: computed at the maximum size, padded for the small non-monotonicities of the cell grid
  h->cap_pyr_block = g.pyr_block;
  h->cap_cand_block = g.cand_block + g.cand_block / 16 + 4096;
  h->cap_cells = g.total_cells + g.total_cells / 8 + 64;
  h->cap_sel_block = g.sel_block;
  h->cap_flist = g.flist_cap;
  h->cap_xtab = 0, h->cap_ytab = 0;
  for (int l = 1; l < g.nlevels; ++l) h->cap_xtab += g.lv[l].pitch + 64, h->cap_ytab += g.lv[l].ph + 8;  // (row tables are padded to groups of 4 rows; smaller images of the same area have other level heights)
  h->cap_oct_tab = 2 * g.nlevels * (cfg->max_width + cfg->max_height + 2) + 64;  // (a level's window is never wider / higher than the image; x 2: other shapes of the same area)
  const size_t B = (size_t)cfg->max_batch;
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) {
    hip_err_set(e, "hipSetDevice");
    delete h;
    return UVO_E_NODEVICE;
  }
#define A(call)                    \
  if ((rc = (call)) != UVO_OK) {   \
    uvo_extractor_destroy(h);      \
    return rc;                     \
  }
  {
    // corner-list regions: sized for both segment heights the launcher may pick, at the maximum resolution
    size_t ce = 0, cn = 0;
    // the segment height depends on the batch size (fast_rows_per_seg): size for the worst batch
    for (int b = 1; b <= (int)B; b = b < 16 ? b + 1 : (int)B) {
      const int rps = fast_rows_per_seg(b);
      const size_t it = (size_t)fast_items_per_frame(g, rps) + 8;
      ce = std::max(ce, (size_t)b * it * (size_t)FS_REGION_ENTRIES);
      cn = std::max(cn, (size_t)b * it);
      if (b == (int)B) break;
    }
    h->cap_cor = ce + ce / 8, h->cap_cor_n = cn + cn / 8 + 64;
    h->cap_flags = (size_t)B * ((size_t)fast_flags_per_frame(g) + fast_flags_per_frame(g) / 8 + 64);
  }
  A(prepare_pyr_tiles(&h->pyr_tiles_max_lds));
  A(alloc_lane(h, 0));
  A(dev_alloc(&h->d_lv, (size_t)kMaxLevels));
  A(dev_alloc(&h->d_cells, (size_t)h->cap_cells));
  A(dev_alloc(&h->d_cell_flag, h->cap_flags / B + 64));
  A(dev_alloc(&h->d_oct_tab, (size_t)h->cap_oct_tab));
  A(dev_alloc(&h->d_ctab, (size_t)h->cap_xtab));
  A(dev_alloc(&h->d_rtab, (size_t)h->cap_ytab));
  A(dev_alloc(&h->d_pattern, (size_t)1024));
  A(dev_alloc(&h->d_patch, (size_t)256));
  // staging for host-buffer calls
  A(dev_alloc(&h->d_imgs, B * (size_t)cfg->max_width * cfg->max_height));
  A(dev_alloc(&h->d_out_kp, B * h->cap_flist));
  A(dev_alloc(&h->d_out_desc, B * h->cap_flist * 32));
  A(dev_alloc(&h->d_n_out, B));
  A(dev_alloc(&h->d_in_kp, B * std::max(cfg->max_input_keypoints, 1)));
  A(dev_alloc(&h->d_n_in, B));
  A(dev_alloc(&h->d_nfn, B));
#undef A
  // circular orientation patch (IC_Angle, src/ORBextractor.cc:125-152): rows v in [-15,15], |u| <= umax[|v|] (749 pixels).
  // k_describe reads it as 31 rows x 8 dwords (u = -16 + 4*chunk + byte); entry row*8 + chunk masks the bytes inside the circle.
  std::vector<uint32_t> patch(256, 0u);
  for (int row = 0; row < 31; ++row) {
    const int v = row - 15, um = h->umax[v < 0 ? -v : v];
    for (int c = 0; c < 8; ++c)
      for (int k = 0; k < 4; ++k) {
        const int u = -16 + 4 * c + k;
        if (u >= -um && u <= um) patch[row * 8 + c] |= 0xffu << (8 * k);
      }
  }
  std::vector<float> patf(1024);
  for (int i = 0; i < 1024; ++i) patf[i] = (float)kPattern[i];
  if (hipMemcpy(h->d_pattern, patf.data(), 4096, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->d_patch, patch.data(), 1024, hipMemcpyHostToDevice) != hipSuccess) {
    uvo_extractor_destroy(h);
    return fail(UVO_E_HIP, "table upload failed");
  }
  *out = h;
  return UVO_OK;
}

void uvo_extractor_destroy(uvo_extractor* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (int i = 0; i < kMaxLanes; ++i)
    if (h->lane[i].stream) (void)hipStreamSynchronize(h->lane[i].stream);
  {
    std::lock_guard<std::mutex> lk(h->followers_mu);
    for (uvo_matcher* m : h->followers) uvo_matcher_orphaned_internal(m);  // their work in these streams is done; they outlive the streams
    h->followers.clear();
  }
  for (int i = 0; i < kMaxLanes; ++i) {
    Lane& L = h->lane[i];
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    L.prof.clear();
    void* lp[] = {L.d_pyr,   L.d_blur,   L.d_cand_xy,   L.d_cand_sc, L.d_pstate, L.d_sel_xy, L.d_sel_sc,
                  L.d_cand_count, L.d_sel_count, L.d_n_final, L.d_cor_n, L.d_cor, L.d_cell_hi, L.d_cand_lo, L.d_cursor, L.d_tpass, L.d_fstat, L.d_fcount, L.d_cell_list, L.d_flist, L.a_imgs, L.a_kp, L.a_desc, L.a_n};
    for (void* p : lp)
      if (p) (void)hipFree(p);
    if (L.a_uploaded) (void)hipEventDestroy(L.a_uploaded);
    for (hipEvent_t& e : L.done)
      if (e) (void)hipEventDestroy(e);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  void* ptrs[] = {h->d_oct_tab, h->d_clahe_lut, h->d_clahe_out, h->d_lv, h->d_cells, h->d_cell_flag, h->d_ctab, h->d_rtab, h->d_pattern, h->d_patch, h->d_imgs, h->d_out_kp,
                  h->d_out_desc, h->d_n_out, h->d_in_kp, h->d_n_in, h->d_nfn, h->d_grid, h->d_grid_score};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->h_pin) (void)hipHostFree(h->h_pin);
  uvo::free_tile_groups(h);
  delete h;
}

int uvo_extractor_levels(const uvo_extractor* h) { return h ? h->cfg.nlevels : UVO_E_BADARG; }
float uvo_extractor_scale_factor(const uvo_extractor* h) { return h ? (float)(double)h->cfg.scale_factor : 0.f; }

int uvo_extractor_tables(const uvo_extractor* h, float* scale, float* inv_scale, int32_t* quota, int32_t* umax16) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  for (int i = 0; i < h->cfg.nlevels; ++i) {
    if (scale) scale[i] = h->scale[i];
    if (inv_scale) inv_scale[i] = h->inv_scale[i];
    if (quota) quota[i] = h->quota[i];
  }
  if (umax16)
    for (int i = 0; i < 16; ++i) umax16[i] = h->umax[i];
  return UVO_OK;
}

int uvo_extract_batch_device(uvo_extractor* h, int batch, const uint8_t* d_imgs, int width, int height, ptrdiff_t stride,
                             ptrdiff_t frame_stride, const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int32_t* d_grid2d, int grid_rows,
                             int grid_cols, int min_px_dist, int full_detect, const int32_t* d_num_feats_needed, uvo_keypoint* d_out_kp,
                             uint8_t* d_out_desc, int cap, int32_t* d_n_out) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  return run_batch_device(h, next_lane(h), batch, d_imgs, width, height, stride, frame_stride, d_in_kp, d_n_in, d_grid2d, grid_rows, grid_cols, min_px_dist,
                          full_detect, d_num_feats_needed, d_out_kp, d_out_desc, cap, d_n_out);
}

// cv::CLAHE::apply parameters (OpenCV 3.4 clahe.cpp): tile size on the extended image, clip limit in pixels, LUT scale
static int clahe_setup(uvo_extractor* h, int batch, int width, int height, double clip_limit, int tiles_x, int tiles_y, int* tile_w, int* tile_h,
                       int* clip, float* lut_scale) {
  if (batch < 1 || batch > h->cfg.max_batch) return fail(UVO_E_BADARG, "batch outside 1..max_batch");
  if (width < 1 || height < 1 || tiles_x < 1 || tiles_y < 1 || tiles_x > width || tiles_y > height) return fail(UVO_E_BADARG, "bad image / tile grid size");
  int ew = width, eh = height;
  if (!(width % tiles_x == 0 && height % tiles_y == 0)) {  // copyMakeBorder(..., 0, tilesY - rows % tilesY, 0, tilesX - cols % tilesX, REFLECT_101)
    ew = width + (tiles_x - width % tiles_x);
    eh = height + (tiles_y - height % tiles_y);
  }
  if (ew - width >= width || eh - height >= height) return fail(UVO_E_BADARG, "tile grid too coarse for REFLECT_101 extension");
  *tile_w = ew / tiles_x, *tile_h = eh / tiles_y;
  const int tileSizeTotal = *tile_w * *tile_h;
  *lut_scale = static_cast<float>(256 - 1) / tileSizeTotal;
  *clip = 0;
  if (clip_limit > 0.0) {
    *clip = static_cast<int>(clip_limit * tileSizeTotal / 256);
    *clip = std::max(*clip, 1);
  }
  const size_t need = (size_t)h->cfg.max_batch * tiles_x * tiles_y * 256;
  if (need > h->clahe_lut_bytes) {
    int rc = sync_all_lanes(h);
    if (rc) return rc;
    if (h->d_clahe_lut) hipFree(h->d_clahe_lut);
    h->d_clahe_lut = nullptr, h->clahe_lut_bytes = 0;
    rc = dev_alloc(&h->d_clahe_lut, need);
    if (rc) return rc;
    h->clahe_lut_bytes = need;
  }
  return UVO_OK;
}

int uvo_clahe_batch_device(uvo_extractor* h, int batch, const uint8_t* d_imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                           double clip_limit, int tiles_x, int tiles_y, uint8_t* d_dst, ptrdiff_t dst_stride, ptrdiff_t dst_frame_stride) {
  if (!h || !d_imgs || !d_dst) return fail(UVO_E_BADARG, "null pointer");
  if (stride < width || dst_stride < width) return fail(UVO_E_BADARG, "stride smaller than the row");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  int tw, th, clip;
  float scale;
  int rc = clahe_setup(h, batch, width, height, clip_limit, tiles_x, tiles_y, &tw, &th, &clip, &scale);
  if (rc) return rc;
  Lane& L = h->lane[next_lane(h)];  // the lane of the extraction that consumes d_dst (same stream: in order behind this kernel)
  {
    Profiler::Scope ps(&L.prof, "k_clahe", L.stream);
    launch_clahe(L.stream, d_imgs, width, height, stride, frame_stride, batch, tiles_x, tiles_y, tw, th, clip, scale, h->d_clahe_lut, d_dst, dst_stride,
                 dst_frame_stride);
  }
  UVO_HIP_CHECK(hipGetLastError());
  return UVO_OK;
}

int uvo_clahe(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t* dst, ptrdiff_t dst_stride) {
  if (!h || !img) return fail(UVO_E_BADARG, "null pointer");
  if (width < 1 || height < 1 || width > h->cfg.max_width || height > h->cfg.max_height || stride < width || (dst && dst_stride < width))
    return fail(UVO_E_BADARG, "image size outside what the handle was sized for");
  if ((int64_t)width * height > (int64_t)h->cfg.max_width * h->cfg.max_height) return fail(UVO_E_BADARG, "image too large");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  if (!h->d_clahe_out) {
    int rc = dev_alloc(&h->d_clahe_out, (size_t)h->cfg.max_width * h->cfg.max_height);
    if (rc) return rc;
  }
  hipStream_t s = h->lane[next_lane(h)].stream;  // the stream uvo_clahe_batch_device() enqueues on
  UVO_HIP_CHECK(hipMemcpy2DAsync(h->d_imgs, width, img, stride, width, (size_t)height, hipMemcpyHostToDevice, s));
  int rc = uvo_clahe_batch_device(h, 1, h->d_imgs, width, height, width, (ptrdiff_t)width * height, clip_limit, tiles_x, tiles_y, h->d_clahe_out, width,
                                  (ptrdiff_t)width * height);
  if (rc) return rc;
  h->clahe_w = width, h->clahe_h = height;
  // dst == NULL: the enhanced image stays in HBM only, for uvo_extract(img = NULL) / uvo_klt_build_pyramid_from_extractor()
  if (dst) UVO_HIP_CHECK(hipMemcpy2DAsync(dst, dst_stride, h->d_clahe_out, width, width, (size_t)height, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));  // the caller's image may be reused
  return UVO_OK;
}

int uvo_extractor_max_keypoints(const uvo_extractor* h) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  return h->cap_flist;
}

int uvo_extractor_synchronize(uvo_extractor* h) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  if (h->spin_wait && h->last_batch > 0 && h->last_batch <= 16) {  // (behind a small batch the wait is short and its wake-up latency counts)
    for (int i = 0; i < kMaxLanes; ++i)
      if (h->lane[i].stream) UVO_HIP_CHECK(wait_stream(h->lane[i].stream, true));
    return UVO_OK;
  }
  return sync_all_lanes(h);
}

int uvo_extractor_tune(uvo_extractor* h, int knob, int value) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  switch (knob) {
    case UVO_TUNE_OCT_WIDE_MAX:
      if (value < 0) return fail(UVO_E_BADARG, "knob value must be >= 0");
      h->oct.wide_max_problems = value;
      return UVO_OK;
    case UVO_TUNE_FAST_MODE: {
      if (value != UVO_FAST_MODE_ADAPTIVE && value != UVO_FAST_MODE_TWO_PASS && value != UVO_FAST_MODE_SINGLE_PASS)
        return fail(UVO_E_BADARG, "UVO_TUNE_FAST_MODE takes UVO_FAST_MODE_ADAPTIVE / _TWO_PASS / _SINGLE_PASS");
      UVO_HIP_CHECK(hipSetDevice(h->device));
      int rc = sync_all_lanes(h);
      if (rc) return rc;
      h->fast_mode = value;
      for (int i = 0; i < kMaxLanes; ++i)
        if (h->lane[i].stream && (rc = set_lane_fast_mode(h, i)) != UVO_OK) return rc;
      return UVO_OK;
    }
    case UVO_TUNE_PYR_FORM:
      if (value < UVO_PYR_FORM_AUTO || value > UVO_PYR_FORM_TILES) return fail(UVO_E_BADARG, "UVO_TUNE_PYR_FORM takes UVO_PYR_FORM_AUTO / _LEVELS / _TILES");
      h->pyr_form = value;
      return UVO_OK;
    case UVO_TUNE_PYR_TILE_GROUP: {
      // value = first level << 16 | tx << 8 | ty (| 1 << 24: 1024-thread workgroups, | 1 << 25: 1024 threads and single-row work items); first level 1
      // starts a new list, 0 returns to the defaults
      const int first = (value >> 16) & 0xff, tx = (value >> 8) & 0xff, ty = value & 0xff;
      if (value != 0 && (first < 1 || first >= kMaxLevels || tx < 1 || ty < 1 || (first > 1 && (h->tile_spec.empty() || first <= (int)((h->tile_spec.back() >> 16) & 0xff)))))
        return fail(UVO_E_BADARG, "UVO_TUNE_PYR_TILE_GROUP takes first << 16 | tx << 8 | ty with ascending first levels starting at 1, or 0");
      UVO_HIP_CHECK(hipSetDevice(h->device));
      int rc = sync_all_lanes(h);
      if (rc) return rc;
      if (value == 0 || first == 1) h->tile_spec.clear();
      if (value != 0) h->tile_spec.push_back((uint32_t)value);
      return h->have_geom ? build_tile_groups(h, h->geom) : UVO_OK;
    }
    case UVO_TUNE_PYR_RING:
      if (value != 0 && value != 4 && value != 8 && value != 12) return fail(UVO_E_BADARG, "UVO_TUNE_PYR_RING takes 0, 4, 8 or 12");
      h->pyr_ring = value;
      return UVO_OK;
    case UVO_TUNE_LEVEL0_INPLACE:
      h->level0_inplace = value != 0;
      return UVO_OK;
    case UVO_TUNE_ZERO_COPY_OUT:
      h->zero_copy_out = value != 0;
      return UVO_OK;
    case UVO_TUNE_SPIN_WAIT:
      h->spin_wait = value != 0;
      return UVO_OK;
    case UVO_TUNE_FEW_FRAMES:
      h->few_frames_shape = value != 0;
      return UVO_OK;
    case UVO_TUNE_FUSE_BLUR_TREE:
      h->fuse_blur_tree = value != 0;
      return UVO_OK;
    case UVO_TUNE_BLUR_ROUNDING:
      if (value != UVO_BLUR_ROUNDING_SCALAR && value != UVO_BLUR_ROUNDING_SSE2) return fail(UVO_E_BADARG, "UVO_TUNE_BLUR_ROUNDING takes UVO_BLUR_ROUNDING_SCALAR / _SSE2");
      h->blur_rounding = value;
      return UVO_OK;
    default:
      return fail(UVO_E_BADARG, "unknown knob");
  }
}

int uvo_extractor_fast_state(uvo_extractor* h, int32_t* pass_threshold, int32_t* fallback_cells, int32_t* cells_per_frame) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  if (!h->have_geom) return fail(UVO_E_BADARG, "no batch has run yet");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  Lane& L = h->lane[h->cur];
  UVO_HIP_CHECK(hipStreamSynchronize(L.stream));
  int32_t t[kMaxLevels], f[kMaxLevels];
  UVO_HIP_CHECK(hipMemcpy(t, L.d_tpass, sizeof(t), hipMemcpyDeviceToHost));
  UVO_HIP_CHECK(hipMemcpy(f, L.d_fstat, sizeof(f), hipMemcpyDeviceToHost));
  for (int l = 0; l < h->geom.nlevels; ++l) {
    if (pass_threshold) pass_threshold[l] = t[l];
    if (fallback_cells) fallback_cells[l] = f[l];
    if (cells_per_frame) cells_per_frame[l] = h->geom.lv[l].n_cells;
  }
  return UVO_OK;
}

int uvo_extractor_set_pipeline(uvo_extractor* h, int depth) {
  if (!h || depth < 1 || depth > kMaxLanes) return fail(UVO_E_BADARG, "pipeline depth must be 1..4");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  int rc = sync_all_lanes(h);
  if (rc) return rc;
  for (int i = 1; i < depth; ++i) {
    rc = alloc_lane(h, i);
    if (rc) return rc;
  }
  h->nlanes = depth;
  if (depth == 1) h->cur = 0;
  return UVO_OK;
}

}  // extern "C"

// build_grid: the occupancy grid is not an input -- it is built on the device from the caller keypoints (src/Tracking.cc:896-912) and
// only returned (grid2d may be NULL)
static int extract_batch_impl(uvo_extractor* h, int batch, const uint8_t* imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                              const uvo_keypoint* in_kp, const int32_t* n_in, int32_t* grid2d, int grid_rows, int grid_cols, int min_px_dist,
                              int full_detect, const int32_t* num_feats_needed, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out,
                              bool build_grid) {
  if (!h || !out_kp || !out_desc || !n_out) return fail(UVO_E_BADARG, "null pointer");
  if (batch < 1 || batch > h->cfg.max_batch) return fail(UVO_E_BADARG, "batch outside 1..max_batch");
  const bool from_clahe = imgs == nullptr;  // the frame is the result of the last uvo_clahe() call, already in HBM
  if (from_clahe) {
    if (batch != 1 || !h->d_clahe_out || h->clahe_w != width || h->clahe_h != height)
      return fail(UVO_E_BADARG, "img == NULL needs a preceding uvo_clahe() of the same size (single frame)");
    stride = width;
  }
  if (width < 1 || height < 1 || width > h->cfg.max_width || height > h->cfg.max_height || stride < width)
    return fail(UVO_E_BADARG, "image size outside what the handle was sized for");
  if ((int64_t)width * height > (int64_t)h->cfg.max_width * h->cfg.max_height) return fail(UVO_E_BADARG, "image too large");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  // uploads, kernels and downloads of this call share one lane: the one the batch is about to run on
  const int li = next_lane(h);
  hipStream_t s = h->lane[li].stream;
  const int in_cap = h->cfg.max_input_keypoints;
  // the reference reads the centre pixel row of a caller keypoint without any bounds check; reject what would
  // leave the padded plane (patch radius 15 + descriptor reach 18 against a 16 px pad)
  if (!full_detect && in_kp && n_in) {
    for (int b = 0; b < batch; ++b) {
      if (n_in[b] < 0 || n_in[b] > in_cap) return fail(UVO_E_BADARG, "n_in outside 0..max_input_keypoints");
      for (int i = 0; i < n_in[b]; ++i) {
        const uvo_keypoint& k = in_kp[(size_t)b * in_cap + i];
        if (build_grid) {
          // the keypoints only mark grid cells, x = (int)(pt.y / d), y = (int)(pt.x / d) (src/Tracking.cc:905-907): the conversion
          // truncates towards zero and the grid has two cells of slack, so a KLT-tracked point a fraction of a cell outside the image
          // marks a cell like any other (the reference accepts it); what the reference would index OUT of its grid is refused here
          const int gx = (int)(k.y / (float)min_px_dist), gy = (int)(k.x / (float)min_px_dist);
          if (!(k.x == k.x && k.y == k.y) || fabsf(k.x) > 1e9f || fabsf(k.y) > 1e9f || gx < 0 || gx >= grid_rows || gy < 0 || gy >= grid_cols)
            return fail(UVO_E_BADARG, "tracked keypoint outside the occupancy grid");
          continue;
        }
        const int cx = (int)lrintf(k.x), cy = (int)lrintf(k.y);
        if (!(cx >= 2 && cx <= width - 3 && cy >= 2 && cy <= height - 3)) return fail(UVO_E_BADARG, "caller keypoint too close to the border");
      }
    }
  }
  // stage inputs (tight rows on the device)
  if (from_clahe) {
    // nothing to upload
  } else if (stride == width && (batch == 1 || frame_stride == (ptrdiff_t)width * height)) {
    UVO_HIP_CHECK(hipMemcpyAsync(h->d_imgs, imgs, (size_t)batch * width * height, hipMemcpyHostToDevice, s));
  } else {
    for (int b = 0; b < batch; ++b)
      UVO_HIP_CHECK(hipMemcpy2DAsync(h->d_imgs + (size_t)b * width * height, width, imgs + (size_t)b * frame_stride, stride, width, (size_t)height,
                                     hipMemcpyHostToDevice, s));
  }
  const uint8_t* d_frames = from_clahe ? h->d_clahe_out : h->d_imgs;
  const bool topup = !full_detect;
  const bool have_in = topup && in_kp && n_in && in_cap > 0;
  // build_grid (uvo_extract_tracked): the caller's keypoints are the TRACKED points, which only fill the occupancy grid -- the extractor
  // itself is called with an empty keypoint vector (`pts0_ext`, src/Tracking.cc:943-946) and returns the new points alone
  const bool describe_in = have_in && !build_grid;
  const int dcap = h->cap_flist;  // device staging capacity per frame
  size_t gb = 0;
  if (topup) {
    if ((!grid2d && !build_grid) || !num_feats_needed) return fail(UVO_E_BADARG, "top-up mode needs grid2d and num_feats_needed");
    gb = (size_t)batch * grid_rows * grid_cols * sizeof(int32_t);
    if (gb > h->grid_bytes) {
      UVO_HIP_CHECK(hipStreamSynchronize(s));
      if (h->d_grid) hipFree(h->d_grid);
      h->d_grid = nullptr;
      int rc = dev_alloc(&h->d_grid, gb / sizeof(int32_t));
      if (rc) return rc;
      h->grid_bytes = gb;
    }
  }
  // pinned bounce buffer: [n_out | nfn | n_in : 3*batch ints][grid][keypoints in, later keypoints + descriptors out]
  const size_t kSmallBatch = 16;  // above this the caller-side copies are a small part of the call
  const bool bounce = (size_t)batch <= kSmallBatch;
  size_t off_grid = 0, off_in = 0, off_kp = 0;
  if (bounce) {
    off_grid = (((size_t)3 * batch * sizeof(int32_t)) + 63) & ~(size_t)63;
    off_in = (off_grid + gb + 63) & ~(size_t)63;  // the caller's keypoints, [frame][in_cap] like the device copy: the kernels may read them here
    off_kp = (off_in + (have_in ? (size_t)batch * in_cap * sizeof(uvo_keypoint) : 0) + 63) & ~(size_t)63;
    const size_t want = off_kp + (size_t)batch * dcap * (sizeof(uvo_keypoint) + 32);
    if (want > h->pin_bytes) {
      UVO_HIP_CHECK(hipStreamSynchronize(s));
      if (h->h_pin) (void)hipHostFree(h->h_pin);
      h->h_pin = nullptr, h->h_pin_dev = nullptr, h->pin_bytes = 0;
      void* p = nullptr;
      if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return fail(UVO_E_NOMEM, "pinned staging allocation failed");
      h->h_pin = (uint8_t*)p, h->pin_bytes = want;
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, p, 0) == hipSuccess) h->h_pin_dev = (uint8_t*)dp;
    }
  }
  int32_t* pin_i = (int32_t*)h->h_pin;
  // Small batches with zero-copy I/O: the kernels read the call's small inputs (feature budget, keypoint counts, the caller's keypoints)
  // where the host put them, in the page-locked region -- each host-to-device copy of a few bytes costs a DMA start-up in front of the
  // first kernel; only the image is uploaded
  uint8_t* const pin_dev = (bounce && h->zero_copy_out) ? h->h_pin_dev : nullptr;
  const int32_t* i_nfn = h->d_nfn;
  const int32_t* i_n_in = h->d_n_in;
  const uvo_keypoint* i_in_kp = h->d_in_kp;
  if (topup) {
    if (bounce) {
      std::memcpy(pin_i + batch, num_feats_needed, sizeof(int32_t) * batch);
      if (pin_dev) i_nfn = (const int32_t*)pin_dev + batch;
      else UVO_HIP_CHECK(hipMemcpyAsync(h->d_nfn, pin_i + batch, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
      if (!build_grid) {
        std::memcpy(h->h_pin + off_grid, grid2d, gb);
        UVO_HIP_CHECK(hipMemcpyAsync(h->d_grid, h->h_pin + off_grid, gb, hipMemcpyHostToDevice, s));
      }
    } else {
      if (!build_grid) UVO_HIP_CHECK(hipMemcpyAsync(h->d_grid, grid2d, gb, hipMemcpyHostToDevice, s));
      UVO_HIP_CHECK(hipMemcpyAsync(h->d_nfn, num_feats_needed, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
    }
    if (have_in) {
      // only the first n_in[b] entries of a frame's slice are ever read on the device
      if (bounce) {
        std::memcpy(pin_i + 2 * batch, n_in, sizeof(int32_t) * batch);
        uvo_keypoint* pk = (uvo_keypoint*)(h->h_pin + off_in);
        for (int b = 0; b < batch; ++b)
          if (n_in[b] > 0) std::memcpy(pk + (size_t)b * in_cap, in_kp + (size_t)b * in_cap, sizeof(uvo_keypoint) * n_in[b]);
        if (pin_dev) {
          i_n_in = (const int32_t*)pin_dev + 2 * batch, i_in_kp = (const uvo_keypoint*)(pin_dev + off_in);
        } else {
          UVO_HIP_CHECK(hipMemcpyAsync(h->d_n_in, pin_i + 2 * batch, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
          for (int b = 0; b < batch; ++b)
            if (n_in[b] > 0)
              UVO_HIP_CHECK(hipMemcpyAsync(h->d_in_kp + (size_t)b * in_cap, pk + (size_t)b * in_cap, sizeof(uvo_keypoint) * n_in[b], hipMemcpyHostToDevice, s));
        }
      } else {
        UVO_HIP_CHECK(hipMemcpyAsync(h->d_n_in, n_in, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
        for (int b = 0; b < batch; ++b)
          if (n_in[b] > 0)
            UVO_HIP_CHECK(hipMemcpyAsync(h->d_in_kp + (size_t)b * in_cap, in_kp + (size_t)b * in_cap, sizeof(uvo_keypoint) * n_in[b],
                                         hipMemcpyHostToDevice, s));
      }
    }
  }
  // build_grid: the occupancy grid is cleared and filled from the tracked keypoints in one launch (no keypoints: cleared)
  if (topup && build_grid) launch_occupancy_grid(s, i_in_kp, have_in ? i_n_in : nullptr, have_in ? in_cap : 0, min_px_dist, grid_rows, grid_cols, h->d_grid, batch);
  // Small batches: k_describe writes the counts, keypoints and descriptors straight into the page-locked bounce region (posted writes over the
  // link: no device-to-host copy -- three DMA start-ups of ~8 us each -- stands between the last kernel and the host)
  uvo_keypoint* const o_kp = pin_dev ? (uvo_keypoint*)(pin_dev + off_kp) : h->d_out_kp;
  uint8_t* const o_desc = pin_dev ? pin_dev + off_kp + (size_t)batch * dcap * sizeof(uvo_keypoint) : h->d_out_desc;
  int32_t* const o_n = pin_dev ? (int32_t*)pin_dev : h->d_n_out;
  int rc = run_batch_device(h, li, batch, d_frames, width, height, width, (ptrdiff_t)width * height, describe_in ? i_in_kp : nullptr,
                            describe_in ? i_n_in : nullptr, topup ? h->d_grid : nullptr, grid_rows, grid_cols, min_px_dist, full_detect,
                            topup ? i_nfn : nullptr, o_kp, o_desc, dcap, o_n);
  if (rc) return rc;
  if (bounce) {
    // The counts, the grid and every frame's whole result slice (dcap records: a frame rarely fills less than 90 % of it) come back in one
    // go -- one synchronisation per call instead of one for the counts and a second for the records they size.
    // The stream is in order: the keypoint uploads above were consumed before the outputs land in the same pinned region.
    uint8_t* const pkp = h->h_pin + off_kp;
    uint8_t* const pde = pkp + (size_t)batch * dcap * sizeof(uvo_keypoint);
    if (topup && grid2d) UVO_HIP_CHECK(hipMemcpyAsync(h->h_pin + off_grid, h->d_grid, gb, hipMemcpyDeviceToHost, s));
    if (!pin_dev) {
      UVO_HIP_CHECK(hipMemcpyAsync(pin_i, h->d_n_out, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, s));
      UVO_HIP_CHECK(hipMemcpyAsync(pkp, h->d_out_kp, (size_t)batch * dcap * sizeof(uvo_keypoint), hipMemcpyDeviceToHost, s));
      UVO_HIP_CHECK(hipMemcpyAsync(pde, h->d_out_desc, (size_t)batch * dcap * 32, hipMemcpyDeviceToHost, s));
    }
    UVO_HIP_CHECK(wait_stream(s, h->spin_wait != 0));
    std::memcpy(n_out, pin_i, sizeof(int32_t) * batch);
    if (topup && grid2d) std::memcpy(grid2d, h->h_pin + off_grid, gb);
    int status = UVO_OK;
    for (int b = 0; b < batch; ++b) {
      int n = n_out[b];
      if (n > cap) {
        status = fail(UVO_E_CAPACITY, "output capacity too small; n_out holds the required size");
        n = cap;
      }
      n = std::min(n, dcap);
      if (n <= 0) continue;
      std::memcpy(out_kp + (size_t)b * cap, pkp + (size_t)b * dcap * sizeof(uvo_keypoint), sizeof(uvo_keypoint) * n);
      std::memcpy(out_desc + (size_t)b * cap * 32, pde + (size_t)b * dcap * 32, (size_t)32 * n);
    }
    return status;
  }
  UVO_HIP_CHECK(hipMemcpyAsync(n_out, h->d_n_out, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  int status = UVO_OK;
  for (int b = 0; b < batch; ++b) {
    int n = n_out[b];
    if (n > cap) {
      status = fail(UVO_E_CAPACITY, "output capacity too small; n_out holds the required size");
      n = cap;
    }
    if (n <= 0) continue;
    UVO_HIP_CHECK(hipMemcpyAsync(out_kp + (size_t)b * cap, h->d_out_kp + (size_t)b * dcap, sizeof(uvo_keypoint) * n, hipMemcpyDeviceToHost, s));
    UVO_HIP_CHECK(hipMemcpyAsync(out_desc + (size_t)b * cap * 32, h->d_out_desc + (size_t)b * dcap * 32, (size_t)32 * n, hipMemcpyDeviceToHost, s));
  }
  if (topup && grid2d) UVO_HIP_CHECK(hipMemcpyAsync(grid2d, h->d_grid, gb, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  return status;
}

extern "C" {

int uvo_extract_batch(uvo_extractor* h, int batch, const uint8_t* imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                      const uvo_keypoint* in_kp, const int32_t* n_in, int32_t* grid2d, int grid_rows, int grid_cols, int min_px_dist,
                      int full_detect, const int32_t* num_feats_needed, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out) {
  return extract_batch_impl(h, batch, imgs, width, height, stride, frame_stride, in_kp, n_in, grid2d, grid_rows, grid_cols, min_px_dist, full_detect,
                            num_feats_needed, out_kp, out_desc, cap, n_out, false);
}

int uvo_extract_tracked(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, const uvo_keypoint* in_kp, int n_in,
                        int min_px_dist, int num_feats_needed, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int* n_out, int32_t* grid2d_out) {
  if (!h || !n_out) return fail(UVO_E_BADARG, "null pointer");
  if (n_in < 0 || n_in > h->cfg.max_input_keypoints || min_px_dist < 1) return fail(UVO_E_BADARG, "n_in outside 0..max_input_keypoints / min_px_dist < 1");
  // Eigen::MatrixXi::Zero((int)(rows / min_px_dist) + 2, (int)(cols / min_px_dist) + 2): src/Tracking.cc:896
  const int grid_rows = height / min_px_dist + 2, grid_cols = width / min_px_dist + 2;
  int32_t nin = n_in, nfn = num_feats_needed, nout = 0;
  int rc = extract_batch_impl(h, 1, img, width, height, stride, (ptrdiff_t)stride * height, in_kp, &nin, grid2d_out, grid_rows, grid_cols, min_px_dist, 0,
                              &nfn, out_kp, out_desc, cap, &nout, true);
  *n_out = nout;
  return rc;
}

int uvo_host_alloc(void** ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail(UVO_E_BADARG, "null pointer / zero size");
  if (hipHostMalloc(ptr, bytes, hipHostMallocPortable) != hipSuccess) return fail(UVO_E_NOMEM, "page-locked allocation failed");
  return UVO_OK;
}
int uvo_host_free(void* ptr) {
  if (ptr && hipHostFree(ptr) != hipSuccess) return fail(UVO_E_HIP, "hipHostFree failed");
  return UVO_OK;
}
int uvo_host_register(void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail(UVO_E_BADARG, "null pointer / zero size");
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
  if (e != hipSuccess) {
    hip_err_set(e, "hipHostRegister");
    return UVO_E_HIP;
  }
  return UVO_OK;
}
int uvo_host_unregister(void* ptr) {
  if (ptr && hipHostUnregister(ptr) != hipSuccess) return fail(UVO_E_HIP, "hipHostUnregister failed");
  return UVO_OK;
}

// The asynchronous host form with its knobs exposed to the sharder (sharder.cpp): only the first n_download frames' results are
// copied to the caller's arrays (the rest of the batch is a halo whose owner downloads it), `after_kernels` (optional) is recorded
// between the kernels and the downloads, and the device-side descriptors / counts of the batch are handed out so that the matcher
// can read them in HBM (valid until the lane is submitted to again).
int uvo_extract_batch_submit_internal(uvo_extractor* h, int batch, int n_download, const uint8_t* imgs, int width, int height, ptrdiff_t stride,
                                      ptrdiff_t frame_stride, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out, int* ticket,
                                      hipEvent_t after_kernels, const uint8_t** d_desc, const int32_t** d_n) {
  if (!h || !imgs || !out_kp || !out_desc || !n_out || !ticket) return fail(UVO_E_BADARG, "null pointer");
  *ticket = -1;
  if (batch < 1 || batch > h->cfg.max_batch || n_download < 0 || n_download > batch) return fail(UVO_E_BADARG, "batch outside 1..max_batch");
  if (width < 1 || height < 1 || width > h->cfg.max_width || height > h->cfg.max_height || stride < width ||
      (int64_t)width * height > (int64_t)h->cfg.max_width * h->cfg.max_height)
    return fail(UVO_E_BADARG, "image size outside what the handle was sized for");
  const int dcap = h->cap_flist;
  if (cap < dcap) return fail(UVO_E_CAPACITY, "cap must be at least uvo_extractor_max_keypoints()");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  int rc = set_geometry(h, width, height);  // before the lane is chosen: a geometry change waits for every lane
  if (rc) return rc;
  const int li = next_lane(h);
  Lane& L = h->lane[li];
  if (L.a_batch) return fail(UVO_E_BADARG, "the next lane still has a batch in flight: wait for it first");
  if (!L.a_imgs) {
    const size_t B = (size_t)h->cfg.max_batch;
    if ((rc = dev_alloc(&L.a_imgs, B * (size_t)h->cfg.max_width * h->cfg.max_height)) || (rc = dev_alloc(&L.a_kp, B * dcap)) ||
        (rc = dev_alloc(&L.a_desc, B * dcap * 32)) || (rc = dev_alloc(&L.a_n, B)))
      return rc;
  }
  hipStream_t s = L.stream;
  if (!L.a_uploaded) UVO_HIP_CHECK(hipEventCreateWithFlags(&L.a_uploaded, hipEventDisableTiming));
  if (h->last_upload) UVO_HIP_CHECK(hipStreamWaitEvent(s, h->last_upload, 0));  // uploads take turns on the link (see uvo_extractor::last_upload)
  if (stride == width && (batch == 1 || frame_stride == (ptrdiff_t)width * height)) {
    UVO_HIP_CHECK(hipMemcpyAsync(L.a_imgs, imgs, (size_t)batch * width * height, hipMemcpyHostToDevice, s));
  } else {
    for (int b = 0; b < batch; ++b)
      UVO_HIP_CHECK(hipMemcpy2DAsync(L.a_imgs + (size_t)b * width * height, width, imgs + (size_t)b * frame_stride, stride, width, (size_t)height,
                                     hipMemcpyHostToDevice, s));
  }
  UVO_HIP_CHECK(hipEventRecord(L.a_uploaded, s));
  h->last_upload = L.a_uploaded;
  const int prev_lane = h->cur;
  rc = run_batch_device(h, li, batch, L.a_imgs, width, height, width, (ptrdiff_t)width * height, nullptr, nullptr, nullptr, 0, 0, 0, 1, nullptr, L.a_kp,
                        L.a_desc, dcap, L.a_n);
  if (rc) {  // no ticket is issued: leave the handle as it was (whatever was enqueued has run out, the lane order is unchanged)
    (void)hipStreamSynchronize(s);
    h->cur = prev_lane;
    return rc;
  }
  if (after_kernels) UVO_HIP_CHECK(hipEventRecord(after_kernels, s));
  // results: whole per-frame slices (a frame holds at most dcap records), frame b lands at b * cap of the caller's arrays
  if (n_download > 0) {
    UVO_HIP_CHECK(hipMemcpyAsync(n_out, L.a_n, sizeof(int32_t) * n_download, hipMemcpyDeviceToHost, s));
    UVO_HIP_CHECK(hipMemcpy2DAsync(out_kp, (size_t)cap * sizeof(uvo_keypoint), L.a_kp, (size_t)dcap * sizeof(uvo_keypoint),
                                   (size_t)dcap * sizeof(uvo_keypoint), (size_t)n_download, hipMemcpyDeviceToHost, s));
    UVO_HIP_CHECK(hipMemcpy2DAsync(out_desc, (size_t)cap * 32, L.a_desc, (size_t)dcap * 32, (size_t)dcap * 32, (size_t)n_download, hipMemcpyDeviceToHost, s));
  }
  L.a_batch = batch;
  *ticket = li;
  if (d_desc) *d_desc = L.a_desc;
  if (d_n) *d_n = L.a_n;
  return UVO_OK;
}

int uvo_extract_batch_submit(uvo_extractor* h, int batch, const uint8_t* imgs, int width, int height, ptrdiff_t stride, ptrdiff_t frame_stride,
                             uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out, int* ticket) {
  return uvo_extract_batch_submit_internal(h, batch, batch, imgs, width, height, stride, frame_stride, out_kp, out_desc, cap, n_out, ticket, nullptr,
                                           nullptr, nullptr);
}

// 1: the lane's batch has delivered everything (uvo_extract_batch_wait would not block), 0: still running, < 0: error
int uvo_extract_batch_done_internal(uvo_extractor* h, int ticket) {
  if (!h || ticket < 0 || ticket >= kMaxLanes || !h->lane[ticket].stream) return fail(UVO_E_BADARG, "bad ticket");
  if (hipSetDevice(h->device) != hipSuccess) return fail(UVO_E_HIP, "hipSetDevice failed");
  const hipError_t e = hipStreamQuery(h->lane[ticket].stream);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  hip_err_set(e, "hipStreamQuery");
  return UVO_E_HIP;
}

int uvo_extract_batch_wait(uvo_extractor* h, int ticket) {
  if (!h || ticket < 0 || ticket >= kMaxLanes || !h->lane[ticket].stream) return fail(UVO_E_BADARG, "bad ticket");
  Lane& L = h->lane[ticket];
  if (!L.a_batch) return fail(UVO_E_BADARG, "no batch in flight on this lane");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  const hipError_t e = hipStreamSynchronize(L.stream);
  L.a_batch = 0;  // the lane is free again whatever the wait reports: a failed batch must not block every later one
  if (e != hipSuccess) {
    hip_err_set(e, "hipStreamSynchronize");
    return UVO_E_HIP;
  }
  return UVO_OK;
}

int uvo_extract(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, const uvo_keypoint* in_kp, int n_in,
                int32_t* grid2d, int grid_rows, int grid_cols, int min_px_dist, int full_detect, int num_feats_needed, uvo_keypoint* out_kp,
                uint8_t* out_desc, int cap, int* n_out) {
  if (!h || !n_out) return fail(UVO_E_BADARG, "null pointer");
  if (n_in < 0 || n_in > h->cfg.max_input_keypoints) return fail(UVO_E_BADARG, "n_in outside 0..max_input_keypoints");
  int32_t nin = n_in, nfn = num_feats_needed, nout = 0;
  const uvo_keypoint* ik = in_kp;
  int rc = uvo_extract_batch(h, 1, img, width, height, stride, (ptrdiff_t)stride * height, ik, &nin, grid2d, grid_rows, grid_cols, min_px_dist,
                             full_detect, &nfn, out_kp, out_desc, cap, &nout);
  *n_out = nout;
  return rc;
}

int uvo_grider_fast(uvo_extractor* h, const uint8_t* img, int width, int height, ptrdiff_t stride, int num_features, int grid_x, int grid_y,
                    int threshold, int nonmax_suppression, uvo_keypoint* out_kp, int cap, int* n_out) {
  if (!h || !img || !out_kp || !n_out) return fail(UVO_E_BADARG, "null pointer");
  *n_out = 0;
  if (width < 7 || height < 7 || width > 4096 || height > 4096 || stride < width || grid_x < 1 || grid_y < 1 || num_features < 0 || cap < 1)
    return fail(UVO_E_BADARG, "bad image size / grid");
  if ((int64_t)width * height > (int64_t)h->cfg.max_width * h->cfg.max_height) return fail(UVO_E_BADARG, "image larger than the handle was sized for");
  const int size_x = width / grid_x, size_y = height / grid_y;
  if (size_x < 1 || size_y < 1) return fail(UVO_E_BADARG, "grid finer than the image (the reference asserts size > 0)");
  const int rois = (width / size_x) * (height / size_y);
  const int keep = num_features / (grid_x * grid_y) + 1;
  Lane& L = h->lane[0];
  if ((size_t)width * height > h->cap_cor || (size_t)rois > h->cap_cor_n)
    return fail(UVO_E_BADARG, "image / grid larger than the handle's scratch");
  const int64_t dcap = (int64_t)h->cfg.max_batch * h->cap_flist;
  if ((int64_t)rois * keep > dcap) return fail(UVO_E_CAPACITY, "num_features + cells exceeds the handle's output staging");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  int rcs = sync_all_lanes(h);
  if (rcs) return rcs;
  if (!h->d_grid_score && (rcs = dev_alloc(&h->d_grid_score, (size_t)h->cfg.max_width * h->cfg.max_height))) return rcs;
  hipStream_t s = L.stream;
  UVO_HIP_CHECK(hipMemcpy2DAsync(h->d_imgs, width, img, stride, width, (size_t)height, hipMemcpyHostToDevice, s));
  UVO_HIP_CHECK(hipMemsetAsync(h->d_n_out, 0, sizeof(int32_t), s));
  launch_grider(s, h->d_imgs, width, height, width, num_features, grid_x, grid_y, threshold, nonmax_suppression ? 1 : 0, h->d_grid_score, L.d_cor,
                L.d_cor_n, h->d_out_kp, (int)std::min<int64_t>(dcap, 1 << 30), h->d_n_out);
  UVO_HIP_CHECK(hipGetLastError());
  int32_t n = 0;
  UVO_HIP_CHECK(hipMemcpyAsync(&n, h->d_n_out, 4, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  *n_out = n;
  int status = UVO_OK;
  if (n > cap) {
    status = fail(UVO_E_CAPACITY, "output capacity too small; n_out holds the required size");
    n = cap;
  }
  if (n > 0) UVO_HIP_CHECK(hipMemcpy(out_kp, h->d_out_kp, sizeof(uvo_keypoint) * n, hipMemcpyDeviceToHost));
  return status;
}

int uvo_extractor_level_dims(const uvo_extractor* h, int level, int* width, int* height) {
  if (!h || !h->have_geom || level < 0 || level >= h->geom.nlevels) return fail(UVO_E_BADARG, "no geometry / bad level");
  *width = h->geom.lv[level].w, *height = h->geom.lv[level].h;
  return UVO_OK;
}

int uvo_extractor_read_plane(uvo_extractor* h, int frame, int level, int which, uint8_t* dst) {
  if (!h || !h->have_geom || level < 0 || level >= h->geom.nlevels || frame < 0 || frame >= h->last_batch || !dst)
    return fail(UVO_E_BADARG, "bad plane request");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  UVO_HIP_CHECK(hipStreamSynchronize(h->lane[h->cur].stream));
  const LevelGeom& L = h->geom.lv[level];
  Lane& LN = h->lane[h->cur];
  if (!which && level >= 1 && LN.ring_used > 0) {
    // the batch wrote the level's ROI and the few pixels around it that a stage reads: this test tap completes the 16-pixel border (the same
    // launch over the whole padded plane; its source -- the ROI of the level below, or the caller's image -- is still there)
    const Geom& g = h->geom;
    Level0View v{nullptr, 0, 0, 0};
    if (level == 1 && LN.l0_src) v = Level0View{LN.l0_src - (int64_t)kPad * LN.l0_stride - kPad, LN.l0_frame_stride, (int)LN.l0_stride, 0};
    launch_resize_level(LN.stream, LN.d_pyr, g.pyr_block, g.lv[level - 1], g.lv[level], h->d_ctab + g.lv[level].xtab_off, h->d_rtab + g.lv[level].ytab_off,
                        h->resize_fast[level], h->last_batch, v, 0);
    UVO_HIP_CHECK(hipStreamSynchronize(LN.stream));
  }
  if (!which && level == 0 && LN.l0_src) {
    // the batch read level 0 in place: this test tap makes the padded plane it never needed (the caller's images must still be there)
    launch_pad_level0(LN.stream, LN.l0_src, h->geom.width, h->geom.height, LN.l0_stride, LN.l0_frame_stride, LN.d_pyr, h->geom.pyr_block, h->geom.lv[0], h->last_batch);
    UVO_HIP_CHECK(hipStreamSynchronize(LN.stream));
  }
  const uint8_t* src = (which ? h->lane[h->cur].d_blur : h->lane[h->cur].d_pyr) + (size_t)frame * h->geom.pyr_block + L.plane_off;
  if (!which) {
    UVO_HIP_CHECK(hipMemcpy2D(dst, L.pw, src, L.pitch, L.pw, L.ph, hipMemcpyDeviceToHost));
    return UVO_OK;
  }
  // the blurred plane lives in HBM as 16 x 8-pixel tiles of 128 bytes (one cache line each: k_describe's windows touch a third of the
  // lines a row-major plane would cost them); pixel (x, y) = tile (y / 8, x / 16), byte (y % 8) * 16 + x % 16
  const size_t bytes = (size_t)L.pitch * ((L.ph + 7) & ~7);
  std::vector<uint8_t> tiled(bytes);
  UVO_HIP_CHECK(hipMemcpy(tiled.data(), src, bytes, hipMemcpyDeviceToHost));
  const int tiles_x = L.pitch >> 4;
  for (int y = 0; y < L.ph; ++y)
    for (int x = 0; x < L.pw; ++x) dst[(size_t)y * L.pw + x] = tiled[((size_t)(y >> 3) * tiles_x + (x >> 4)) * 128 + (y & 7) * 16 + (x & 15)];
  return UVO_OK;
}

int uvo_extractor_read_candidates(uvo_extractor* h, int frame, int level, int32_t* dst_xys, int cap, int* n) {
  if (!h || !h->have_geom || level < 0 || level >= h->geom.nlevels || frame < 0 || frame >= h->last_batch || !n)
    return fail(UVO_E_BADARG, "bad candidate request");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  UVO_HIP_CHECK(hipStreamSynchronize(h->lane[h->cur].stream));
  const LevelGeom& L = h->geom.lv[level];
  int32_t cnt = 0;
  UVO_HIP_CHECK(hipMemcpy(&cnt, h->lane[h->cur].d_cand_count + (size_t)frame * h->geom.nlevels + level, 4, hipMemcpyDeviceToHost));
  *n = cnt;
  cnt = std::min(cnt, L.cand_cap);
  const int m = std::min(cnt, cap);
  if (m > 0 && dst_xys) {
    std::vector<uint32_t> xy(m), sc(m);
    const size_t off = (size_t)frame * h->geom.cand_block + L.cand_off;
    UVO_HIP_CHECK(hipMemcpy(xy.data(), h->lane[h->cur].d_cand_xy + off, (size_t)4 * m, hipMemcpyDeviceToHost));
    UVO_HIP_CHECK(hipMemcpy(sc.data(), h->lane[h->cur].d_cand_sc + off, (size_t)4 * m, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; ++i) {
      dst_xys[3 * i] = (int32_t)(xy[i] & 0xffff);
      dst_xys[3 * i + 1] = (int32_t)(xy[i] >> 16);
      dst_xys[3 * i + 2] = (int32_t)sc[i];
    }
  }
  return UVO_OK;
}

int uvo_extractor_profile(uvo_extractor* h, int enable) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  int rc = sync_all_lanes(h);
  if (rc) return rc;
  for (int i = 0; i < kMaxLanes; ++i) {
    h->lane[i].prof.on = enable != 0;
    h->lane[i].prof.clear();
  }
  return UVO_OK;
}

int uvo_extractor_profile_only(uvo_extractor* h, const char* kernel_name) {
  if (!h) return fail(UVO_E_BADARG, "null handle");
  for (int i = 0; i < kMaxLanes; ++i) h->lane[i].prof.only = kernel_name ? kernel_name : "";
  return UVO_OK;
}

int uvo_extractor_kernel_times(uvo_extractor* h, char* names, int names_cap, float* ms, int32_t* launches, int cap, int* n) {
  if (!h || !names || !ms || !launches || !n) return fail(UVO_E_BADARG, "null pointer");
  UVO_HIP_CHECK(hipSetDevice(h->device));
  int rc = sync_all_lanes(h);
  if (rc) return rc;
  // fold lane 1's records into lane 0's report
  for (int i = 1; i < kMaxLanes; ++i) {
    for (auto& r : h->lane[i].prof.recs) h->lane[0].prof.recs.push_back(r);
    h->lane[i].prof.recs.clear();
  }
  *n = h->lane[0].prof.report(names, names_cap, ms, launches, cap);
  return UVO_OK;
}

hipStream_t uvo_extractor_stream_internal(uvo_extractor* h) { return h->lane[h->cur].stream; }
void uvo_extractor_add_follower_internal(uvo_extractor* h, uvo_matcher* m) {
  std::lock_guard<std::mutex> lk(h->followers_mu);
  if (std::find(h->followers.begin(), h->followers.end(), m) == h->followers.end()) h->followers.push_back(m);
}
void uvo_extractor_drop_follower_internal(uvo_extractor* h, uvo_matcher* m) {
  std::lock_guard<std::mutex> lk(h->followers_mu);
  h->followers.erase(std::remove(h->followers.begin(), h->followers.end(), m), h->followers.end());
}
int uvo_extractor_device_internal(uvo_extractor* h) { return h->device; }
int uvo_extractor_next_lane_internal(const uvo_extractor* h) { return next_lane(h); }
const uint8_t* uvo_extractor_clahe_internal(uvo_extractor* h, int* width, int* height) {
  *width = h->clahe_w, *height = h->clahe_h;
  return h->d_clahe_out;
}

}  // extern "C"
