// Host execution of the fused pyramid launch's tile plan (u-vip-slam_amd/csrc/pyr_tiles.hpp) -- test infrastructure, not product code.
// The levels are cut into groups (one launch each); every workgroup of a group's plan is run in turn the way k_pyr_tiles runs it (the
// group's levels in order, level l from the level l - 1 tile the SAME workgroup kept in its "LDS", the group's first level from memory:
// the image or the plane an earlier group stored), with an independent per-pixel restatement of cv::resize's 8-bit
// INTER_LINEAR arithmetic instead of the kernel's packed form.  What it proves about a plan:
//   * every byte of a level's written area (ROI + ring) is stored by exactly one workgroup;
//   * a workgroup never reads an LDS byte it has not written (tap of non-zero weight), every 12-byte tap window stays inside the tile's
//     allocation, every LDS tile inside the plan's LDS size, the two LDS regions of consecutive levels do not overlap;
//   * the planes equal whatever the caller compares them with (the oracle's ComputePyramid).
// Returns 0, or a positive code naming the violated property.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
using std::size_t;

#include "../../u-vip-slam_amd/csrc/pyr_tiles.hpp"

// group_first[g] = first level of group g (group_first[0] = 1, ascending); tx / ty per group
extern "C" int emu_pyr_tiles_run(const uint8_t* img, int stride, int nlevels, const int* lw, const int* lh, int ring, int ngroups, const int* group_first, const int* gtx,
                                 const int* gty, uint32_t max_lds, uint8_t* planes, int64_t* plane_off, int* pitch_out, int64_t* stats) {
  using namespace uvo;
  if (nlevels < 2 || nlevels > 16) return 1;
  std::vector<PyrTileDims> dims(nlevels);
  int64_t off = 0;
  for (int l = 0; l < nlevels; ++l) {
    dims[l] = PyrTileDims{lw[l], lh[l], (lw[l] + 2 * kPyrTilePad + 63) / 64 * 64};
    plane_off[l] = off, pitch_out[l] = dims[l].pitch;
    off += (int64_t)dims[l].pitch * (lh[l] + 2 * kPyrTilePad);
  }
  std::vector<std::vector<ResizeCol>> ct(nlevels);
  std::vector<std::vector<ResizeRow>> rt(nlevels);
  std::vector<const ResizeCol*> cp(nlevels, nullptr);
  std::vector<const ResizeRow*> rp(nlevels, nullptr);
  for (int l = 1; l < nlevels; ++l) {
    int ok = 0;
    pyr_build_level_tables(lw[l - 1], lh[l - 1], lw[l], lh[l], dims[l].pitch, ct[l], rt[l], &ok);
    if (!ok) return 2;
    cp[l] = ct[l].data(), rp[l] = rt[l].data();
  }
  std::vector<std::vector<uint8_t>> stored(nlevels);
  for (int l = 1; l < nlevels; ++l) stored[l].assign((size_t)dims[l].pitch * (lh[l] + 2 * kPyrTilePad), 0);
  stats[0] = stats[1] = stats[2] = 0;
  if (ngroups < 1 || group_first[0] != 1) return 1;
  for (int g = 0; g < ngroups; ++g) {
  const int first = group_first[g], last = g + 1 < ngroups ? group_first[g + 1] - 1 : nlevels - 1, tx = gtx[g], ty = gty[g];
  PyrTilePlan P;
  if (!pyr_tile_plan_build(dims.data(), nlevels, first, last, cp.data(), rp.data(), ring, tx, ty, max_lds, P)) return 3;
  stats[0] += P.computed_px, stats[1] += P.owned_px, stats[2] = stats[2] > P.lds_bytes ? stats[2] : P.lds_bytes;
  std::vector<uint8_t> lds(P.lds_bytes + 16), valid(P.lds_bytes + 16);
  const int ntiles = tx * ty;
  for (int t = 0; t < ntiles; ++t) {
    std::fill(valid.begin(), valid.end(), (uint8_t)0);
    std::fill(lds.begin(), lds.end(), (uint8_t)0xA5);
    for (int l = first; l <= last; ++l) {
      const PyrTileLevel& T = P.lv[(size_t)t * nlevels + l];
      const PyrTileLevel& S = P.lv[(size_t)t * nlevels + l - 1];  // the tile this level reads (l > first)
      const int spitch = S.lw + kPyrTileSlack, dpitch = T.lw + kPyrTileSlack;
      if (T.lw > 0) {
        if (T.lx0 % 4 || T.lw % 4 || T.lrows <= 0) return 4;
        if ((uint64_t)T.lds_off + (uint64_t)dpitch * T.lrows > P.lds_bytes) return 5;
        if (l > first && S.lw > 0) {  // the tile being read and the tile being written must not overlap
          const uint64_t a0 = S.lds_off, a1 = a0 + (uint64_t)spitch * S.lrows, b0 = T.lds_off, b1 = b0 + (uint64_t)dpitch * T.lrows;
          if (a0 < b1 && b0 < a1) return 6;
        }
        for (int y = 0; y < T.lrows; ++y) std::memset(&valid[T.lds_off + (size_t)y * dpitch], 0, (size_t)dpitch);  // a reused region holds an older level
      }
      if (T.ncw <= 0 || T.nrg <= 0) continue;
      if (T.cy0 % 4) return 7;
      if (l > first) {  // the level's coefficient tables are staged in LDS behind every tile of the workgroup, 16-byte aligned, inside the allocation
        const uint64_t t0 = T.tab_off, t1 = t0 + (uint64_t)(T.ncw + T.nrg) * 32;
        if (t0 % 16 || t1 > P.lds_bytes) return 14;
        for (int m = first; m <= last; ++m) {
          const PyrTileLevel& M = P.lv[(size_t)t * nlevels + m];
          if (M.lw > 0 && t0 < (uint64_t)M.lds_off + (uint64_t)(M.lw + kPyrTileSlack) * M.lrows) return 15;
          if (m > first && m != l) {
            const uint64_t m0 = M.tab_off, m1 = m0 + (uint64_t)(M.ncw + M.nrg) * 32;
            if (t0 < m1 && m0 < t1) return 16;
          }
        }
      }
      const int sw = lw[l - 1], dw = lw[l], dh = lh[l];
      const int x_hi = (kPyrTilePad + dw + ring + 3) / 4 * 4;
      for (int rg = 0; rg < T.nrg; ++rg)
        for (int c = 0; c < T.ncw; ++c) {
          const int wx = T.cx0w + c;
          const ResizeCol* e = cp[l] + 4 * wx;
          const int base = e[0].pad;
          if (l > first) {  // the kernel reads the 12-byte window [base, base + 12) of both source rows from the tile
            if (base < S.lx0 || base + 12 > S.lx0 + spitch) return 8;
          }
          for (int j = 0; j < kPyrTileRows; ++j) {
            const int py = T.cy0 + rg * kPyrTileRows + j;
            if (py >= (int)rt[l].size()) return 9;
            const ResizeRow& r = rp[l][py];
            uint8_t out[4];
            for (int i = 0; i < 4; ++i) {
              const int sx0 = e[i].sx, sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
              const uint32_t a0 = e[i].a0 >> 4, a1 = e[i].a1 >> 4;  // the 11-bit weights
              uint32_t acc[2];
              for (int k = 0; k < 2; ++k) {
                const int sy = k ? r.sy1 : r.sy0;
                uint32_t L, R;
                if (l == 1) {
                  L = img[(size_t)sy * stride + sx0], R = img[(size_t)sy * stride + sx1];
                } else if (l == first) {  // the plane an earlier group stored: ROI pixels only
                  const uint8_t* sp = planes + plane_off[l - 1] + (size_t)(sy + kPyrTilePad) * dims[l - 1].pitch + kPyrTilePad;
                  L = sp[sx0], R = sp[sx1];
                } else {
                  if (sy < S.ly0 || sy >= S.ly0 + S.lrows) return 10;
                  const size_t o0 = S.lds_off + (size_t)(sy - S.ly0) * spitch + (sx0 - S.lx0), o1 = S.lds_off + (size_t)(sy - S.ly0) * spitch + (sx1 - S.lx0);
                  if (sx0 < S.lx0 || sx0 >= S.lx0 + S.lw || !valid[o0]) return 11;
                  if (a1 && (sx1 >= S.lx0 + S.lw || !valid[o1])) return 12;
                  L = lds[o0], R = a1 ? lds[o1] : 0;
                }
                acc[k] = L * a0 + R * a1;  // horizontal pass: 11-bit fixed point
              }
              // vertical pass of the 8-bit generic path: ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
              out[i] = (uint8_t)(((((uint32_t)r.b0 * (acc[0] >> 4)) >> 16) + (((uint32_t)r.b1 * (acc[1] >> 4)) >> 16) + 2) >> 2);
            }
            // HBM: the owned cell
            if (wx >= T.ox0w && wx < T.ox1w && py >= T.oy0 && py < T.oy1) {
              if (wx * 4 < kPyrTilePad - ring || wx * 4 + 4 > x_hi || py < kPyrTilePad - ring || py >= kPyrTilePad + dh + ring) return 13;
              uint8_t* d = planes + plane_off[l] + (size_t)py * dims[l].pitch + wx * 4;
              for (int i = 0; i < 4; ++i) d[i] = out[i], ++stored[l][(size_t)py * dims[l].pitch + wx * 4 + i];
            }
            // LDS: what level l + 1 reads
            if (T.lw > 0) {
              const int x = wx * 4 - kPyrTilePad - T.lx0, y = py - kPyrTilePad - T.ly0;
              if (x >= 0 && x < T.lw && y >= 0 && y < T.lrows)
                for (int i = 0; i < 4; ++i) lds[T.lds_off + (size_t)y * dpitch + x + i] = out[i], valid[T.lds_off + (size_t)y * dpitch + x + i] = 1;
            }
          }
        }
    }
  }
  }
  for (int l = 1; l < nlevels; ++l) {
    const int x_hi = (kPyrTilePad + lw[l] + ring + 3) / 4 * 4;
    for (int py = 0; py < lh[l] + 2 * kPyrTilePad; ++py)
      for (int px = 0; px < dims[l].pitch; ++px) {
        const bool in = px >= kPyrTilePad - ring && px < x_hi && py >= kPyrTilePad - ring && py < kPyrTilePad + lh[l] + ring;
        if (stored[l][(size_t)py * dims[l].pitch + px] != (in ? 1 : 0)) return 20 + (in ? 0 : 1);
      }
  }
  return 0;
}
