// Schedule of the fused pyramid kernel (k_pyramid, pyramid.hip), compiled on the host per (geometry, band count).
// Replaces the launch chain of ORBextractor::ComputePyramid (src/ORBextractor.cc:963-1004): copyMakeBorder at level 0 (:996) and, per
// level, cv::resize(level l-1 -> l) + copyMakeBorder(REFLECT_101 | ISOLATED) (:982,:988) -- one dependent chain of seven resizes.
//
// The chain is kept inside ONE launch: a workgroup owns a (frame, band) -- a band is a range of rows of every level -- and walks it in
// macro-steps separated by a workgroup barrier.  In a step, level 0 copies its next R0 image rows into the padded plane, and every
// level l >= 1 produces the rows whose source rows (level l-1) were written by this workgroup in EARLIER steps: a row written before
// a barrier is visible to the whole workgroup behind it (same CU, same L1 / L2), no agent-scope traffic, and the rows in flight stay
// in the L2 they were written through.  Level l therefore runs l steps behind level 0.
// ROLES: the columns of every level are cut into 64-lane chunks (lane = 4 output bytes; level 0: 16 bytes), and every (level, chunk)
// is a role that ONE wavefront keeps for the whole band: its column tables and the horizontal pass of the last source row it has seen
// stay in registers, so every source row is fetched and interpolated horizontally exactly once per role (1.2 per output row at scale
// 1.2 where a per-pixel kernel interpolates 2).  What a step does is a table: per (step, level) the NEW source rows the level's roles
// stream over and, per source row, the output row it completes (weights, store offsets) -- the kernel's step is straight-line code.
// Bands: each band computes what it owns plus the few rows of the levels below that its own rows need (recomputed, bit-identical to
// the neighbour's copy -- both store the same bytes), so bands never wait for one another.
//
// Free of HIP headers: tests/emu/pyr_schedule_emu.cpp executes the schedule on the host (plain C++) and checks that every row is
// read only after the step that wrote it, and that the planes equal the oracle's.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace uvo {

constexpr int kPyrMaxLevels = 16;
constexpr int kPyrMaxSrcRows = 7;   // new source rows a role streams over per step (all loaded up front: 7 x 3 dwords per lane)
constexpr int kPyrMaxStepRows = 8;  // output rows of a level per step: one per new source row + the clamped last row
constexpr int kPyrMinSrcRows = 6;   // a level waits for this many new source rows before it runs (a step costs a role the same whatever it streams)
constexpr int kPyrNop = 255;
constexpr int kPyrPad = 16;         // EDGE_THRESHOLD, src/ORBextractor.cc:78
#ifndef UVO_PYR_PITCH_ALIGN
#define UVO_PYR_PITCH_ALIGN 64
#endif
// Row pitch of the padded planes.  With 128 (whole cache lines: a line never holds bytes of two rows) the fused launch reads a row back
// 30 % sooner -- a line shared with the row below, written a step later, is completed from HBM before the L2 serves it -- but the planes
// grow by 7 % and every other stage pays for that: 269 000 instead of 278 000 frames/s (profiles/r04_pyramid_ab.txt).
constexpr int kPyrPitchAlign = UVO_PYR_PITCH_ALIGN;

constexpr int32_t kPyrNoStore = 0x40000000;  // a row offset outside every plane: the buffer hardware drops the store

// What a level does in a step, 160 bytes per (step, level).  The level's roles stream over the NEW source rows k_lo .. k_lo + nsrc - 1;
// slot j (0..6) describes the output row that is emitted while source row k_lo + j passes by (its lower tap): the vertical weights and
// where the row is stored (kPyrNoStore: no row; `dual` = the REFLECT_101 pad row that repeats it, rows 1..16 and h-17..h-2 of a level).
// Slot 7 is the level's clamped LAST row when both its taps are the last source row (emitted after the stream with that row twice).
// The kernel's step is straight-line code over these words: no per-row control flow.
struct PyrStepLevel {
  int16_t k_lo, nsrc;   // level >= 1: new source rows; level 0: image rows [k_lo, k_lo + nsrc) to copy
  int16_t lo, hi;       // output ROI rows [lo, hi) emitted in this step (the host-side emulation checks the slots against them)
  uint32_t flags;       // bit 0: slot 0's row has both taps on source row k_lo (the level's clamped FIRST row); bit 1: slot 7 is in use
  uint32_t pad;
  float b[8][2];        // ibeta / 65536 of the slot's row (upper tap, lower tap)
  int32_t soff[8];      // byte offset of the slot's row in the padded plane, or kPyrNoStore
  int32_t dual[8];      // byte offset of the pad row that repeats it, or kPyrNoStore
  int32_t pad2[4];
};
static_assert(sizeof(PyrStepLevel) == 160, "the kernel reads it as ten 16-byte words");

struct PyrRole {  // 4 bytes; slot s of wavefront w = roles[w * nslots + s]
  uint8_t level;   // 0: copy image rows into the padded level-0 plane; l >= 1: resize level l-1 -> l; kPyrNop: slot unused
  uint8_t pad;
  uint16_t chunk;  // lanes cover 64 consecutive units from 64 * chunk (units: 16-byte groups of a padded row at level 0, dwords above)
};

struct PyrRow {  // per ROI output row of a level >= 1 (cv::resize's vertical tables: SURVEY.md A.2)
  int16_t sy0, sy1;  // the two source rows, clamped into the source level
  float b0, b1;      // ibeta / 65536: the vertical pass is floor(q0 * b0) + floor(q1 * b1) on the fp32 pipe (pyramid.hip)
};

struct PyrDims {
  int w, h, pitch;
};

struct PyrSchedule {
  int nbands = 0, nwaves = 0, nslots = 0, nlevels = 0;
  std::vector<PyrRole> roles;          // [nwaves * nslots], the same for every band
  std::vector<PyrStepLevel> steps;     // [nsteps][nlevels]
  std::vector<int32_t> band_step;      // [nbands + 1] first step of a band
  // statistics for the tests / DESIGN
  int64_t rows_computed = 0, rows_owned = 0;
  int nroles = 0, max_wave_load = 0, sum_wave_load = 0;
};

// cv::resize INTER_LINEAR row table of one level pair exactly as resizeGeneric_ builds it, per ROI output row
inline void pyr_build_rows(int sh, int dh, std::vector<PyrRow>& rows) {
  rows.resize(dh);
  const double scale_y = 1. / ((double)dh / sh);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)fy;
    sy -= (sy > fy);  // cvFloor
    fy -= sy;
    const int sy0 = std::min(std::max(sy, 0), sh - 1), sy1 = std::min(std::max(sy + 1, 0), sh - 1);
    const int b0 = (int)lrintf((1.f - fy) * 2048.f), b1 = (int)lrintf(fy * 2048.f);  // saturate_cast<short>(cvRound)
    rows[dy] = PyrRow{(int16_t)sy0, (int16_t)sy1, (float)b0 * (1.0f / 65536.0f), (float)b1 * (1.0f / 65536.0f)};
  }
}

inline int pyr_chunks(const PyrDims& d, int level) { return ((level == 0 ? d.pitch / 16 : d.pitch / 4) + 63) / 64; }
// (resize roles, copy roles) of the levels first_level .. nlevels - 1 of a geometry
inline void pyr_role_count(const PyrDims* lv, int first_level, int nlevels, int& nresize, int& ncopy) {
  nresize = 0, ncopy = first_level == 0 ? pyr_chunks(lv[0], 0) : 0;
  for (int l = std::max(first_level, 1); l < nlevels; ++l) nresize += pyr_chunks(lv[l], l);
}

// Roles onto wavefronts.  Copy roles (level 0) get wavefronts of their own, one each -- their loop has nothing in common with a resize
// role's -- and the resize roles are dealt heaviest first onto the least loaded of the other wavefronts that still has a free slot
// (weight = rows of the level).
inline bool pyr_assign_roles(const PyrDims* lv, int first_level, int nlevels, int nwaves, int nslots, PyrSchedule& S) {
  struct R {
    int level, chunk, weight;
  };
  std::vector<R> all;
  for (int l = std::max(first_level, 1); l < nlevels; ++l)
    for (int c = 0; c < pyr_chunks(lv[l], l); ++c) all.push_back(R{l, c, lv[l].h});
  const int ncopy = first_level == 0 ? pyr_chunks(lv[0], 0) : 0, nres = nwaves - ncopy;
  S.nroles = (int)all.size() + ncopy;
  if (nres < (all.empty() ? 0 : 1) || (int)all.size() > nres * nslots) return false;
  std::stable_sort(all.begin(), all.end(), [](const R& a, const R& b) { return a.weight > b.weight; });
  std::vector<int> load(nwaves, 0), used(nwaves, 0);
  S.roles.assign((size_t)nwaves * nslots, PyrRole{(uint8_t)kPyrNop, 0, 0});
  for (const R& r : all) {
    int best = -1;
    for (int w = 0; w < nres; ++w)
      if (used[w] < nslots && (best < 0 || load[w] < load[best])) best = w;
    if (best < 0) return false;
    S.roles[(size_t)best * nslots + used[best]++] = PyrRole{(uint8_t)r.level, 0, (uint16_t)r.chunk};
    load[best] += r.weight;
  }
  for (int c = 0; c < ncopy; ++c) S.roles[(size_t)(nres + c) * nslots] = PyrRole{0, 0, (uint16_t)c}, load[nres + c] = (lv[0].h + 3) / 4;
  S.max_wave_load = *std::max_element(load.begin(), load.end());
  S.sum_wave_load = 0;
  for (int x : load) S.sum_wave_load += x;
  return true;
}

inline void pyr_blank_step(PyrStepLevel& T) {
  T.k_lo = 0, T.nsrc = 0, T.lo = 0, T.hi = 0, T.flags = 0, T.pad = 0;
  for (int j = 0; j < 8; ++j) T.b[j][0] = T.b[j][1] = 0.f, T.soff[j] = T.dual[j] = kPyrNoStore;
  for (int j = 0; j < 4; ++j) T.pad2[j] = 0;
}

// Fills T's slots for a stream over the source rows k_lo .. k_hi of a level: every output row from `a` on (below y_end) whose lower tap
// is among them, in the slot of that source row.  Returns the first row not emitted, or -1 when the rows cannot be expressed (a row whose
// lower tap has passed already, taps that are no neighbours, ...).  d = the level, sh = rows of the level below.
inline int pyr_fill_slots(const std::vector<PyrRow>& rows, const PyrDims& d, int sh, int k_lo, int k_hi, int a, int y_end, PyrStepLevel& T) {
  const int pitch = d.pitch, h = d.h;
  T.k_lo = (int16_t)k_lo, T.nsrc = (int16_t)(k_hi - k_lo + 1), T.lo = (int16_t)a;
  int y = a;
  while (y < y_end && (int)rows[y].sy1 <= k_hi) {
    const PyrRow& r = rows[y];
    const int j = (int)r.sy1 - k_lo;
    if (j < 0 || (r.sy0 != r.sy1 && r.sy0 != r.sy1 - 1)) return -1;  // emitted while its lower tap passes by; taps are neighbours
    int dual = kPyrNoStore;
    if (y >= 1 && y <= kPyrPad) dual = (kPyrPad - y) * pitch;                                 // top pad: row -y = row y
    else if (y >= h - 1 - kPyrPad && y <= h - 2) dual = (kPyrPad + 2 * (h - 1) - y) * pitch;  // bottom pad
    int slot = j;
    if (r.sy0 == r.sy1) {
      // both taps on one row: the level's first row (slot 0 of the stream that starts the level), or its last row (after the stream)
      if (y == 0 && j == 0 && k_lo == 0) T.flags |= 1u;
      else if (y == h - 1 && (int)r.sy1 == sh - 1 && j == T.nsrc - 1) slot = 7, T.flags |= 2u;
      else return -1;
    } else if (j == 0 && (int)r.sy0 != k_lo - 1) return -1;  // the upper tap of slot 0 is the row carried into the stream
    if (T.soff[slot] != kPyrNoStore) return -1;  // one row per slot
    T.b[slot][0] = r.b0, T.b[slot][1] = r.b1, T.soff[slot] = (kPyrPad + y) * pitch, T.dual[slot] = dual;
    ++y;
  }
  T.hi = (int16_t)y;
  return y;
}

// The streaming form (k_pyr_stream): a level cut into static blocks of kPyrMaxSrcRows source rows; block i streams over source rows
// 7 i .. 7 i + 6 and emits the rows whose lower tap they are.  A wavefront walks a run of consecutive blocks of one column chunk; it
// computes the row in front of its first block itself (the upper tap of that block's slot 0).
inline bool pyr_build_blocks(const std::vector<PyrRow>& rows, const PyrDims& d, int sh, std::vector<PyrStepLevel>& out) {
  out.clear();
  int y = 0;
  for (int k_lo = 0; k_lo < sh; k_lo += kPyrMaxSrcRows) {
    PyrStepLevel T;
    pyr_blank_step(T);
    const int k_hi = std::min(k_lo + kPyrMaxSrcRows, sh) - 1;
    y = pyr_fill_slots(rows, d, sh, k_lo, k_hi, y, d.h, T);
    if (y < 0) return false;
    out.push_back(T);
  }
  return y == d.h;
}

// Builds the schedule.  rows[l] = row table of level l (l >= 1).  r0 = level-0 rows copied per step (<= kPyrMaxSrcRows).
// first_level: the levels below it exist already (written by earlier launches); first_level = 0 builds the whole pyramid from the image.
// Returns false when the roles do not fit nwaves x nslots, or the geometry cannot be scheduled (never for scale factors >= 1).
inline bool pyr_build_schedule(const PyrDims* lv, int first_level, int nlevels, const std::vector<PyrRow>* rows, int nbands, int nwaves, int nslots, int r0,
                               PyrSchedule& S) {
  S = PyrSchedule();
  S.nbands = nbands, S.nwaves = nwaves, S.nslots = nslots, S.nlevels = nlevels;
  if (r0 < 1 || r0 > kPyrMaxSrcRows || nlevels < 1 || nlevels > kPyrMaxLevels || first_level < 0 || first_level >= nlevels) return false;
  if (!pyr_assign_roles(lv, first_level, nlevels, nwaves, nslots, S)) return false;
  S.band_step.push_back(0);
  for (int band = 0; band < nbands; ++band) {
    // rows this band owns, and rows it computes (owned + what its rows of the level below need), top-down
    int lo[kPyrMaxLevels], hi[kPyrMaxLevels];
    for (int l = nlevels - 1; l >= 0; --l) {
      if (l < first_level) {  // complete before the launch
        lo[l] = 0, hi[l] = lv[l].h;
        continue;
      }
      lo[l] = (int)((int64_t)band * lv[l].h / nbands), hi[l] = (int)((int64_t)(band + 1) * lv[l].h / nbands);
      S.rows_owned += hi[l] - lo[l];
      if (l + 1 < nlevels && hi[l + 1] > lo[l + 1]) {
        const int need_lo = rows[l + 1][lo[l + 1]].sy0, need_hi = rows[l + 1][hi[l + 1] - 1].sy1 + 1;  // sy0 / sy1 are monotone in the row
        if (hi[l] <= lo[l]) lo[l] = need_lo, hi[l] = need_hi;
        lo[l] = std::min(lo[l], need_lo), hi[l] = std::max(hi[l], need_hi);
      }
      S.rows_computed += hi[l] - lo[l];
    }
    int done[kPyrMaxLevels], cons[kPyrMaxLevels];  // next output row of each level; next unseen source row of each level >= 1
    for (int l = 0; l < nlevels; ++l) {
      done[l] = l < first_level ? hi[l] : lo[l];
      cons[l] = (l > 0 && l >= first_level && hi[l] > lo[l]) ? rows[l][lo[l]].sy0 : 0;
    }
    for (int guard = 0;; ++guard) {
      bool all = true;
      for (int l = 0; l < nlevels; ++l) all = all && done[l] >= hi[l];
      if (all) break;
      if (guard > 100000) return false;
      int avail[kPyrMaxLevels];  // rows written before this step's barrier
      for (int l = 0; l < nlevels; ++l) avail[l] = done[l];
      bool progress = false;
      PyrStepLevel blank;
      pyr_blank_step(blank);
      std::vector<PyrStepLevel> st(nlevels, blank);
      if (done[0] < hi[0]) {
        const int a = done[0], b = std::min(a + r0, hi[0]);
        st[0].k_lo = (int16_t)a, st[0].nsrc = (int16_t)(b - a), st[0].lo = (int16_t)a, st[0].hi = (int16_t)b;
        done[0] = b, progress = true;
      }
      for (int l = std::max(first_level, 1); l < nlevels; ++l) {
        if (done[l] >= hi[l]) continue;
        // the level's roles stream over the source rows written so far (as many as the step's register budget takes) and emit every
        // output row whose lower tap is among them; a source row that completes no output row yet is carried as the next row's upper tap
        const int a = done[l], k_lo = cons[l];
        // (never past the lower tap of the band's last row: the level below may be complete, the band's share of it is not all of it)
        const int k_hi = std::min(std::min(avail[l - 1] - 1, k_lo + kPyrMaxSrcRows - 1), (int)rows[l][hi[l] - 1].sy1);
        if (k_hi < k_lo) continue;
        // a step costs a role the same whether it streams two rows or seven: wait until the level below has delivered a good batch
        // (or has nothing more to deliver)
        if (k_hi - k_lo + 1 < std::min(kPyrMinSrcRows, r0) && k_hi < (int)rows[l][hi[l] - 1].sy1 && done[l - 1] < hi[l - 1] && avail[l - 1] < hi[l - 1]) continue;
        const int y = pyr_fill_slots(rows[l], lv[l], lv[l - 1].h, k_lo, k_hi, a, hi[l], st[l]);
        if (y < 0 || (y > a && k_lo < lo[l - 1])) return false;
        cons[l] = k_hi + 1, done[l] = y, progress = true;
      }
      if (!progress) return false;  // cannot happen while a level below still has rows to deliver
      S.steps.insert(S.steps.end(), st.begin(), st.end());
    }
    S.band_step.push_back((int32_t)(S.steps.size() / nlevels));
  }
  return true;
}

}  // namespace uvo
