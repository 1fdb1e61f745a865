// Host execution of the fused pyramid kernel's schedule (u-vip-slam_amd/csrc/pyr_schedule.hpp), compiled with a plain C++ compiler.
// Test infrastructure: it runs the step table exactly as k_pyramid's workgroups would -- band by band, step by step, every role on
// its 64-unit column chunk -- with an independent integer restatement of cv::resize's two passes, and checks the
// discipline the kernel relies on:
//   * a role reads a source row only if THIS band wrote all of it in an EARLIER step (rows written in the same step are behind no barrier);
//   * the new source rows of a level follow one another without gap or repeat (a role sees every source row exactly once), at most
//     kPyrMaxSrcRows per step; every output row is emitted in the step in which its lower tap passes by, in the slot of that source row
//     (the clamped last row in slot 7), its upper tap the slot before (slot 0: the row carried from the step before) -- the slots'
//     weights, store offsets and flags are re-derived here from cv::resize's row table and must equal the schedule's words;
//   * every (level, chunk) is the role of exactly one wavefront slot, copy roles sit alone in slot 0 of their wavefront;
//   * every byte of every padded plane is written (by at least one band), and bands that write the same byte write the same value.
// Returns 0 and the planes (the caller compares them with the oracle's pyramid), or a positive code naming the violated property.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../u-vip-slam_amd/csrc/pyr_schedule.hpp"

namespace {
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}
inline int cv_floor(float v) {
  int i = (int)v;
  return i - (i > v);
}
}  // namespace

extern "C" int emu_pyr_schedule_run(const uint8_t* img, int stride, int first_level, int nlevels, const int* lw, const int* lh, int nbands, int nwaves, int nslots, int r0,
                                    uint8_t* planes /* concatenated padded planes, pitch = (w + 32) rounded up to kPyrPitchAlign */, int64_t* plane_off, int* pitch_out,
                                    int* stats /* [0] steps of the longest band, [1] roles, [2] heaviest wavefront (rows), [3] rows computed, [4] rows owned, [5] sum of wavefront loads */) {
  using namespace uvo;
  PyrDims dims[kPyrMaxLevels];
  std::vector<PyrRow> rows[kPyrMaxLevels];
  int64_t off = 0;
  for (int l = 0; l < nlevels; ++l) {
    dims[l] = PyrDims{lw[l], lh[l], (lw[l] + 2 * kPyrPad + kPyrPitchAlign - 1) / kPyrPitchAlign * kPyrPitchAlign};
    pitch_out[l] = dims[l].pitch;
    plane_off[l] = off;
    off += (int64_t)dims[l].pitch * (lh[l] + 2 * kPyrPad);
    if (l > 0) pyr_build_rows(lh[l - 1], lh[l], rows[l]);
  }
  PyrSchedule S;
  if (!pyr_build_schedule(dims, first_level, nlevels, rows, nbands, nwaves, nslots, r0, S)) return 1;
  if ((int)S.band_step.size() != nbands + 1 || (int)S.roles.size() != nwaves * nslots) return 2;
  // every (level, chunk) exactly once; copy roles alone in their wavefront
  {
    std::vector<std::vector<int>> seen(nlevels);
    for (int l = 0; l < nlevels; ++l) seen[l].assign(pyr_chunks(dims[l], l), 0);
    for (int w = 0; w < nwaves; ++w)
      for (int k = 0; k < nslots; ++k) {
        const PyrRole& r = S.roles[(size_t)w * nslots + k];
        if (r.level == kPyrNop) continue;
        if (r.level >= nlevels || r.chunk >= seen[r.level].size()) return 3;
        ++seen[r.level][r.chunk];
        if (r.level == 0 && k != 0) return 3;
        if (r.level != 0 && S.roles[(size_t)w * nslots].level == 0) return 3;
      }
    for (int l = 0; l < nlevels; ++l)
      for (int c : seen[l])
        if (c != (l >= first_level ? 1 : 0)) return 3;  // the levels below first_level belong to the streaming launches
  }
  // horizontal tables per level (independent of the device's packed format): xofs, ialpha as resizeGeneric_ builds them
  std::vector<int> xofs[kPyrMaxLevels];
  std::vector<short> ialpha[kPyrMaxLevels];
  for (int l = 1; l < nlevels; ++l) {
    const int sw = lw[l - 1], dw = lw[l];
    const double scale_x = 1. / ((double)dw / sw);
    xofs[l].resize(dw), ialpha[l].resize(2 * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = cv_floor(fx);
      fx -= sx;
      if (sx < 0) fx = 0, sx = 0;
      if (sx >= sw - 1) fx = 0, sx = sw - 1;
      xofs[l][dx] = sx;
      ialpha[l][2 * dx] = (short)lrintf((1.f - fx) * 2048.f), ialpha[l][2 * dx + 1] = (short)lrintf(fx * 2048.f);
    }
  }
  const int64_t total = off;
  std::vector<uint8_t> written(total, 0);  // by any band
  std::memset(planes, 0xCD, total);
  int longest = 0;
  if ((int)S.steps.size() != S.band_step.back() * nlevels) return 2;
  // one (step or block, level) entry: checks its words against cv::resize's row table, computes its rows, stores them.
  // row_step: per level the step that wrote each padded row of THIS band (-1: not yet); next_k / next_y: the stream state of the level.
  auto process = [&](const PyrStepLevel& T, int l, int s, std::vector<std::vector<int>>& row_step, std::vector<int>& next_k, std::vector<int>& next_y) -> int {
    if (T.nsrc <= 0) {
      for (int j = 0; j < 8; ++j)
        if (T.soff[j] != kPyrNoStore || T.dual[j] != kPyrNoStore) return 4;
      return 0;
    }
    if (T.nsrc > kPyrMaxSrcRows) return 6;
    const int pitch = dims[l].pitch, h = lh[l], w = lw[l];
    int y0, y1;  // output rows of this entry
    if (l == 0) {
      y0 = T.k_lo, y1 = T.k_lo + T.nsrc;
      if (y0 < 0 || y1 > h) return 4;
    } else {
      const int k_lo = T.k_lo, k_hi = T.k_lo + T.nsrc - 1;
      if (k_lo < 0 || k_hi >= lh[l - 1]) return 6;
      if (next_k[l] >= 0 && k_lo != next_k[l]) return 6;  // every source row passes a role exactly once, in order
      // the new source rows were written by this band in an earlier step (the carried row was read in the step before)
      for (int k = k_lo; k <= k_hi; ++k)
        if (row_step[l - 1][kPyrPad + k] < -1 || row_step[l - 1][kPyrPad + k] >= s) return 9;
      // the rows this entry must emit: every row not yet emitted whose lower tap is among the new rows -- re-derived here, slot by slot
      y0 = next_y[l] >= 0 ? next_y[l] : T.lo;
      y1 = y0;
      float eb[8][2];
      int esoff[8], edual[8];
      uint32_t eflags = 0;
      for (int q = 0; q < 8; ++q) eb[q][0] = eb[q][1] = 0.f, esoff[q] = edual[q] = kPyrNoStore;
      while (y1 < h && rows[l][y1].sy1 <= k_hi) {
        const PyrRow& R = rows[l][y1];
        if (R.sy1 < k_lo) return 6;  // its lower tap has passed already: the row was left behind
        int slot = R.sy1 - k_lo;
        if (R.sy0 == R.sy1) {
          if (y1 == 0 && slot == 0) eflags |= 1u;
          else if (y1 == h - 1 && R.sy1 == k_hi) slot = 7, eflags |= 2u;
          else return 8;
        } else {
          if (R.sy0 != R.sy1 - 1) return 8;
          if (slot == 0 && k_lo == 0) return 6;  // the upper tap of slot 0 is the row in front of the stream: there is none in front of row 0
        }
        if (esoff[slot] != kPyrNoStore) return 7;  // two rows in one slot
        eb[slot][0] = R.b0, eb[slot][1] = R.b1, esoff[slot] = (kPyrPad + y1) * pitch;
        if (y1 >= 1 && y1 <= kPyrPad) edual[slot] = (kPyrPad - y1) * pitch;
        else if (y1 >= h - 1 - kPyrPad && y1 <= h - 2) edual[slot] = (kPyrPad + 2 * (h - 1) - y1) * pitch;
        ++y1;
      }
      if (T.lo != y0 || T.hi != y1 || T.flags != eflags) return 10;
      for (int q = 0; q < 8; ++q)
        if (T.soff[q] != esoff[q] || T.dual[q] != edual[q] || (esoff[q] != kPyrNoStore && (T.b[q][0] != eb[q][0] || T.b[q][1] != eb[q][1]))) return 11;
      next_k[l] = k_hi + 1;
    }
    next_y[l] = y1;
    uint8_t* P = planes + plane_off[l];
    uint8_t* Wr = written.data() + plane_off[l];
    std::vector<uint8_t> v(pitch);
    for (int y = y0; y < y1; ++y) {
      if (l == 0) {
        for (int x = 0; x < pitch; ++x) v[x] = img[(int64_t)y * stride + reflect101(x - kPyrPad, w)];  // columns past the padded width (pitch slack) hold something valid too
      } else {
        const PyrRow& R = rows[l][y];
        const uint8_t* S0 = planes + plane_off[l - 1] + (int64_t)(kPyrPad + R.sy0) * dims[l - 1].pitch + kPyrPad;
        const uint8_t* S1 = planes + plane_off[l - 1] + (int64_t)(kPyrPad + R.sy1) * dims[l - 1].pitch + kPyrPad;
        const int b0 = (int)lrintf(R.b0 * 65536.f), b1 = (int)lrintf(R.b1 * 65536.f);
        const int sw = lw[l - 1];
        for (int x = 0; x < pitch; ++x) {
          const int dx = reflect101(x - kPyrPad, w);
          const int sx = xofs[l][dx], a0 = ialpha[l][2 * dx], a1 = ialpha[l][2 * dx + 1];
          const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
          const int h0 = S0[sx] * a0 + S0[sx1] * a1, h1 = S1[sx] * a0 + S1[sx1] * a1;
          v[x] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
      }
      auto store_row = [&](int prow) -> int {
        for (int x = 0; x < pitch; ++x) {
          if (Wr[(int64_t)prow * pitch + x] && P[(int64_t)prow * pitch + x] != v[x]) return 20;  // two writers disagree
          P[(int64_t)prow * pitch + x] = v[x], Wr[(int64_t)prow * pitch + x] = 1;
        }
        if (row_step[l][prow] < 0) row_step[l][prow] = s;
        return 0;
      };
      int rc = store_row(kPyrPad + y);
      if (rc) return rc;
      if (y >= 1 && y <= kPyrPad && (rc = store_row(kPyrPad - y))) return rc;
      if (y >= h - 1 - kPyrPad && y <= h - 2 && (rc = store_row(kPyrPad + 2 * (h - 1) - y))) return rc;
    }
    return 0;
  };
  // ---- the streaming launches: levels below first_level, one after another (every launch finds the level below complete) ----
  {
    std::vector<std::vector<int>> row_step(nlevels);
    for (int l = 0; l < nlevels; ++l) row_step[l].assign(lh[l] + 2 * kPyrPad, -2);
    std::vector<int> next_k(nlevels, -1), next_y(nlevels, -1);
    for (int l = 0; l < first_level; ++l) {
      if (l == 0) {
        PyrStepLevel T;
        pyr_blank_step(T);
        for (int y = 0; y < lh[0]; y += kPyrMaxSrcRows) {
          T.k_lo = (int16_t)y, T.nsrc = (int16_t)(y + kPyrMaxSrcRows <= lh[0] ? kPyrMaxSrcRows : lh[0] - y);
          int rc = process(T, 0, l, row_step, next_k, next_y);
          if (rc) return rc;
        }
      } else {
        std::vector<PyrStepLevel> blocks;
        if (!pyr_build_blocks(rows[l], dims[l], lh[l - 1], blocks)) return 12;
        if ((int)blocks.size() != (lh[l - 1] + kPyrMaxSrcRows - 1) / kPyrMaxSrcRows) return 12;
        for (size_t b = 0; b < blocks.size(); ++b) {
          if (blocks[b].k_lo != (int)b * kPyrMaxSrcRows) return 12;  // a wavefront finds block b's rows without looking anything up
          int rc = process(blocks[b], l, l, row_step, next_k, next_y);
          if (rc) return rc;
        }
        if (next_y[l] != lh[l]) return 12;
      }
    }
  }
  for (int band = 0; band < nbands; ++band) {
    // per band: the step that wrote each padded row (a level's roles all write it in the same step), -1 = not yet; the levels of the
    // streaming launches were complete before the launch (-1 counts as "before every step" for them)
    std::vector<std::vector<int>> row_step(nlevels);
    for (int l = 0; l < nlevels; ++l) row_step[l].assign(lh[l] + 2 * kPyrPad, l < first_level ? -1 : -2);
    std::vector<int> next_k(nlevels, -1), next_y(nlevels, -1);  // per level: the next unseen source row / output row (-1: the level has not started)
    const int s0 = S.band_step[band], s1 = S.band_step[band + 1];
    longest = s1 - s0 > longest ? s1 - s0 : longest;
    for (int s = s0; s < s1; ++s)
      for (int l = 0; l < nlevels; ++l) {
        const PyrStepLevel& T = S.steps[(size_t)s * nlevels + l];
        if (l < first_level && T.nsrc != 0) return 13;
        int rc = process(T, l, s, row_step, next_k, next_y);
        if (rc) return rc;
      }
  }
  for (int64_t i = 0; i < total; ++i)
    if (!written[i]) return 30;  // a byte no band wrote
  stats[0] = longest, stats[1] = S.nroles, stats[2] = S.max_wave_load, stats[3] = (int)S.rows_computed, stats[4] = (int)S.rows_owned, stats[5] = S.sum_wave_load;
  return 0;
}
