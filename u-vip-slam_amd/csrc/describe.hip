// Output assembly, orientation and descriptors.
//   k_assemble : builds the per-frame final slot list in the reference's output order
//                (ORBextractor::operator() filter block, src/ORBextractor.cc:861-913, + ComputeKeyPointsCopy :523-534)
//   k_describe : one wavefront per final slot -- IC_Angle (:125-152) on the un-blurred level, steered rBRIEF
//                (computeOrbDescriptor :156-195) on the blurred level, keypoint record assembly (:820-830, :948-959)
#include "common.hpp"
#include "uvo_math.hpp"

namespace uvo {

// ---------------------------------------------------------------------------------------------------------
// One workgroup per frame.
// FullDetect: allKeypoints = allKeypointsTemp (:911) -> plain concatenation of the per-level survivor lists.
// Top-up    : level-0 list starts with the caller's keypoints (:863); then levels are walked 0..L-1 in list order,
//             a point is skipped when its min_px_dist cell of grid_2d is occupied, else accepted and the cell
//             incremented; per-level cap num_featsneeded*(8-level)/30 with a counter that carries across levels
//             (:878,:892-897), global cap num_featsneeded (:898-901).  Order dependent -> one thread walks it.
// TOPUP = false is the FullDetect concatenation alone: no LDS, so its workgroups find room next to the other lane's kernels
// The adaptive FAST mode of a pipeline lane (fast.hip, octree.hip): k_octree left every (frame, level) problem's count of fall-back
// cells -- cells without a keypoint at fastTh, src/ORBextractor.cc:795 -- in fa.fcount; workgroup 0 of k_assemble sums the batch and
// sets the threshold each level streams at in the lane's NEXT batch: above 22 % fall-back cells one pass at 7 with the per-cell vote,
// below 14 % the two-pass form (measured break-even: a fall-back cell costs 5.5 x what the two-pass form saves per cell, 18 %).
// Both forms give the same candidates.  Stream order makes the new values visible to this lane's next k_fast_score; no other lane
// reads them.
// the ONE place a level's next streaming threshold is decided from its batch's count of fall-back cells (called by adapt_fast_mode for
// large batches and by k_describe<DIRECT> for the frame or two of a latency call)
__device__ __forceinline__ void decide_fast_pass(const LevelGeom* __restrict__ lv, int l, int batch, int64_t zc, const FastAdapt& fa) {
  const int64_t cells = (int64_t)lv[l].n_cells * batch;
  fa.last[l] = (int32_t)zc;
  if (fa.adapt && fa.fast_th > 7 && cells > 0) {
    const int cur_t = fa.tpass[l];
    if (cur_t > 7 && zc * 100 > cells * 22)
      fa.tpass[l] = 7;
    else if (cur_t <= 7 && zc * 100 < cells * 14)
      fa.tpass[l] = fa.fast_th;
  }
}

__device__ __forceinline__ void adapt_fast_mode(const LevelGeom* __restrict__ lv, int nlevels, int batch, const FastAdapt& fa) {
  __shared__ int s_sum[kMaxLevels];
  if (threadIdx.x < kMaxLevels) s_sum[threadIdx.x] = 0;
  __syncthreads();
  for (int l = 0; l < nlevels; ++l) {
    int z = 0;
    for (int f = threadIdx.x; f < batch; f += blockDim.x) z += fa.fcount[f * nlevels + l];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) z += __shfl_xor(z, o, 64);
    if ((threadIdx.x & 63) == 0 && z) atomicAdd(&s_sum[l], z);
  }
  __syncthreads();
  if ((int)threadIdx.x < nlevels) decide_fast_pass(lv, (int)threadIdx.x, batch, s_sum[threadIdx.x], fa);
}

template <bool TOPUP>
__global__ __launch_bounds__(256) void k_assemble(const LevelGeom* __restrict__ lv, int nlevels, FastAdapt fa, const uint32_t* __restrict__ sel_xy,
                                                  const uint32_t* __restrict__ sel_sc, int sel_block,
                                                  const int32_t* __restrict__ sel_count, const int32_t* __restrict__ n_in, int in_cap,
                                                  int32_t* __restrict__ grid, int grid_rows, int grid_cols, int min_px_dist,
                                                  int full_detect, const int32_t* __restrict__ nfn, FinalSlot* __restrict__ flist,
                                                  int flist_cap, int32_t* __restrict__ n_final) {
  const int f = blockIdx.x;
  if (f == 0) adapt_fast_mode(lv, nlevels, (int)gridDim.x, fa);
  const uint32_t* sxy = sel_xy + (int64_t)f * sel_block;
  const uint32_t* ssc = sel_sc + (int64_t)f * sel_block;
  const int32_t* cnt = sel_count + f * nlevels;
  FinalSlot* out = flist + (int64_t)f * flist_cap;
  if (!TOPUP || full_detect) {
    int base = 0;
    for (int l = 0; l < nlevels; ++l) {
      int n = cnt[l];
      n = n > lv[l].sel_cap ? lv[l].sel_cap : n;
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t xy = sxy[lv[l].sel_off + i];
        FinalSlot s;
        s.x = (float)((int)(xy & 0xffff) + kMinBorder);
        s.y = (float)((int)(xy >> 16) + kMinBorder);
        s.level = l;
        s.aux = (int32_t)ssc[lv[l].sel_off + i];
        if (base + i < flist_cap) out[base + i] = s;
      }
      base += n;
    }
    if (threadIdx.x == 0) n_final[f] = base;
    return;
  }
  if constexpr (TOPUP) {
  // top-up mode
  const int nin = n_in ? min(n_in[f], in_cap) : 0;
  for (int i = threadIdx.x; i < nin; i += blockDim.x) {
    FinalSlot s;
    s.x = 0.f, s.y = 0.f;  // coordinates are read from in_kp by k_describe
    s.level = (int32_t)0x80000000;
    s.aux = i;
    if (i < flist_cap) out[i] = s;
  }
  // The walk is sequential (a point's fate depends on the cells taken by the points before it), so its memory has to be close:
  // the survivor lists and the occupancy grid are staged in LDS by the whole workgroup, one thread walks them there, and the
  // grid goes back in parallel.  Frames whose lists or grid exceed the staging arrays walk in HBM.
  constexpr int AS_PTS = 2560, AS_GRID = 6144;
  __shared__ uint32_t s_xy[AS_PTS];
  __shared__ int s_lbase[kMaxLevels + 1];
  int32_t* Gm = grid + (int64_t)f * grid_rows * grid_cols;
  const int gsize = grid_rows * grid_cols;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int l = 0; l < nlevels; ++l) {
      s_lbase[l] = run;
      int n = cnt[l];
      run += n > lv[l].sel_cap ? lv[l].sel_cap : n;
    }
    s_lbase[nlevels] = run;
  }
  __syncthreads();
  const bool staged = s_lbase[nlevels] <= AS_PTS && gsize <= AS_GRID;
  if (staged) {
    for (int l = 0; l < nlevels; ++l) {
      const int n = s_lbase[l + 1] - s_lbase[l];
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        s_xy[s_lbase[l] + i] = sxy[lv[l].sel_off + i];
      }
    }
  }
  const int need = nfn[f];
  if (staged) {
    // Level by level (the levels are truly sequential), inside a level in parallel: a point is accepted iff its cell is free
    // and it is the first point of the level in that cell -- and it lies before the point at which the reference's loop leaves
    // the level, which has a closed form: the k-th accepted point has KP_counter = carry + k and Total_counter = total + k, the
    // loop breaks at the first k with carry + k == numofpoint (level cap, counter reset; checked first) or total + k == need
    // (global cap, break_key).
    // s_first[cell]: -1 = occupied, else the lowest index of a point of the current level that falls into the (free) cell
    __shared__ int s_first[AS_GRID];
    __shared__ int s_cell[AS_PTS];
    __shared__ int s_wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_in_block();
    for (int i = tid; i < gsize; i += blockDim.x) s_first[i] = Gm[i] > 0 ? -1 : 0x7fffffff;
    __syncthreads();
    int pos = nin, Total_counter = 0, KP_counter = 0;
    bool break_key = false;
    for (int l = 0; l < nlevels && !break_key; ++l) {
      const int n = s_lbase[l + 1] - s_lbase[l], lb = s_lbase[l];
      if (n == 0) continue;
      const int numofpoint = need * (8 - l) / 30;
      const float scale = lv[l].scale;
      for (int i = tid; i < n; i += blockDim.x) {
        const uint32_t xy = s_xy[lb + i];
        const float px = (float)((int)(xy & 0xffff) + kMinBorder), py = (float)((int)(xy >> 16) + kMinBorder);
        const float tx = px * scale, ty = py * scale;
        const int r = (int)(ty / (float)min_px_dist), c = (int)(tx / (float)min_px_dist);
        const int cell = c * grid_rows + r;  // Eigen column-major
        s_cell[i] = cell;
        atomicMin(&s_first[cell], i);  // stays -1 for an occupied cell
      }
      __syncthreads();
      const int kcap1 = numofpoint - KP_counter >= 1 ? numofpoint - KP_counter : 0x7fffffff;
      const int kcap2 = need - Total_counter >= 1 ? need - Total_counter : 0x7fffffff;
      const int kcap = kcap1 < kcap2 ? kcap1 : kcap2;
      int nfirst = 0;  // would-be accepted points seen so far (uniform)
      for (int i0 = 0; i0 < n && nfirst < kcap; i0 += blockDim.x) {
        const int i = i0 + tid;
        const bool first = i < n && s_first[s_cell[i]] == i;
        const uint64_t m = __ballot(first);
        if (lane == 0) s_wcnt[wv] = (int)__popcll(m);
        __syncthreads();
        int rank = nfirst + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        for (int q = 0; q < wv; ++q) rank += s_wcnt[q];
        const int round_total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        if (first && rank < kcap) {
          const uint32_t xy = s_xy[lb + i];
          FinalSlot fs;
          fs.x = (float)((int)(xy & 0xffff) + kMinBorder), fs.y = (float)((int)(xy >> 16) + kMinBorder), fs.level = l;
          fs.aux = (int32_t)ssc[lv[l].sel_off + i];
          if (pos + rank < flist_cap) out[pos + rank] = fs;
          Gm[s_cell[i]] += 1;       // grid_2d(r, c)++ : one writer per cell (the accepted points of a level lie in distinct cells)
          s_first[s_cell[i]] = -1;  // occupied from now on (no other point compares equal to -1)
        }
        nfirst += round_total;
        __syncthreads();
      }
      const int take = nfirst < kcap ? nfirst : kcap;
      pos += take;
      Total_counter += take;
      if (take == kcap1) {
        KP_counter = 0;  // level cap reached: `break` out of the level, the global cap is not looked at for this point
      } else if (take == kcap2) {
        break_key = true;
      } else {
        KP_counter += take;
      }
      for (int i = tid; i < n; i += blockDim.x)
        if (s_first[s_cell[i]] != -1) s_first[s_cell[i]] = 0x7fffffff;  // cells that stayed free (or lost their point to the cap)
      __syncthreads();
    }
    if (tid == 0) n_final[f] = pos;
  } else if (threadIdx.x == 0) {
    int pos = nin;
    int Total_counter = 0, KP_counter = 0;
    bool break_key = false;
    for (int l = 0; l < nlevels && !break_key; ++l) {
      const int n = s_lbase[l + 1] - s_lbase[l];
      if (n == 0) continue;
      const int numofpoint = need * (8 - l) / 30;
      const float scale = lv[l].scale;
      for (int i = 0; i < n; ++i) {
        const uint32_t xy = sxy[lv[l].sel_off + i];
        const float px = (float)((int)(xy & 0xffff) + kMinBorder), py = (float)((int)(xy >> 16) + kMinBorder);
        const float tx = px * scale, ty = py * scale;
        const int r = (int)(ty / (float)min_px_dist), c = (int)(tx / (float)min_px_dist);
        int32_t* cell = &Gm[(int64_t)c * grid_rows + r];  // Eigen column-major
        if (*cell > 0) continue;
        FinalSlot fs;
        fs.x = px, fs.y = py, fs.level = l, fs.aux = (int32_t)ssc[lv[l].sel_off + i];
        if (pos < flist_cap) out[pos] = fs;
        ++pos;
        *cell += 1;
        ++KP_counter;
        ++Total_counter;
        if (KP_counter == numofpoint) {
          KP_counter = 0;
          break;
        }
        if (Total_counter == need) {
          break_key = true;
          break;
        }
      }
    }
    n_final[f] = pos;
  }
  }  // TOPUP
}

// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ballot_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// 4 wavefronts per workgroup, DK_PER_WAVE consecutive final slots per wavefront.  The stages of a keypoint are a chain of
// dependent gathers (slot -> level geometry -> patch pixels -> angle -> steered sample points -> blurred pixels), so a
// wavefront keeps several keypoints in flight: all patch loads are issued before the first reduction, all blurred-pixel
// loads before the first ballot, and the pattern table is read once for the group.
constexpr int DK_PER_WAVE = 4;
// The blurred plane is tiled (gauss.hip): 16 x 8-pixel tiles of one 128-byte line.  A keypoint's window (u in [-18, 21], v in [-18, 18])
// lies inside a grid of at most 4 x 6 tiles; the tiles are copied whole -- one dwordx4 per lane, eight lanes per line -- into a
// row-major LDS window of 48 rows x 64 bytes, from which the sample points are read as before.
constexpr int DW_TX = 4, DW_TY = 6, DW_PITCH = DW_TX * 16, DW_ROWS = DW_TY * 8, DW_DWORDS = DW_ROWS * DW_PITCH / 4;

#ifndef UVO_OCC_DESCRIBE
#define UVO_OCC_DESCRIBE 1  // more workgroups per CU change nothing here (measured)
#endif
#ifndef UVO_DESC_WAVES
#define UVO_DESC_WAVES 2   // wavefronts per workgroup: 24.6 KB of LDS -- fits beside four k_fast_score workgroups of the other pipeline lane (a
                           // four-wavefront workgroup's 49 KB only beside three): +2 % frames/s at 640x512, +-0 at 1920x1080
#endif
constexpr int DK_WAVES = UVO_DESC_WAVES;
// DIRECT (FullDetect, a frame or two): there is no k_assemble launch in front -- the final list of a FullDetect frame is the levels' quad-tree
// survivors one after the other (src/ORBextractor.cc:915-960), so a wavefront finds its slots' (level, index) from the eight survivor counts
// itself: one stage less in a chain of latency-bound stages.  (With a batch that fills the chip the slot arithmetic in front of the dependent
// fetches costs k_describe more than the launch it saves: measured, DESIGN.md.)
struct DirectSel {
  const uint32_t* sel_xy;
  const uint32_t* sel_sc;
  const int32_t* sel_count;
  int sel_block;
  FastAdapt fa;           // the lane's adaptive FAST mode: what k_assemble's workgroup 0 does otherwise
  int batch;
};
template <bool DIRECT>
__global__ __launch_bounds__(64 * DK_WAVES, UVO_OCC_DESCRIBE) void k_describe(const LevelGeom* __restrict__ lv, int nlevels, const uint8_t* __restrict__ pyr,
                                                  const uint8_t* __restrict__ blur, int64_t pyr_block,
                                                  const FinalSlot* __restrict__ flist, int flist_cap, const int32_t* __restrict__ n_final,
                                                  const uvo_keypoint* __restrict__ in_kp, int in_cap, const float* __restrict__ pattern,
                                                  const uint32_t* __restrict__ patch, uvo_keypoint* __restrict__ out_kp,
                                                  uint8_t* __restrict__ out_desc, int cap, int32_t* __restrict__ n_out, Level0View l0, DirectSel ds) {
  // a frame's keypoints stay on one XCD: their 37-row windows overlap heavily (1000 windows cover a level about once), and with
  // round-robin placement every XCD's L2 would fetch the same lines
  const int vb = xcd_contiguous((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int bx = vb % (int)gridDim.x, f = vb / (int)gridDim.x;
  const int lane = threadIdx.x & 63;
  const int slot0 = (bx * DK_WAVES + wave_in_block()) * DK_PER_WAVE;
  int n;
  int lcount[kMaxLevels];  // DIRECT: the levels' survivor counts (wave-uniform)
  if constexpr (DIRECT) {
    n = 0;
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) {
      int c = 0;
      if (l < nlevels) {
        c = ds.sel_count[f * nlevels + l];
        c = c > lv[l].sel_cap ? lv[l].sel_cap : c;
      }
      lcount[l] = c, n += c;
    }
    if (bx == 0 && f == 0 && (int)threadIdx.x < nlevels) {  // the lane's fall-back statistics and pass thresholds (adapt_fast_mode, for a frame or two)
      const int l = (int)threadIdx.x;
      int64_t zc = 0;
      for (int b = 0; b < ds.batch; ++b) zc += ds.fa.fcount[b * nlevels + l];
      decide_fast_pass(lv, l, ds.batch, zc, ds.fa);
    }
  } else {
    n = n_final[f];
  }
  if (bx == 0 && threadIdx.x == 0) n_out[f] = n;
  n = n > flist_cap ? flist_cap : n;
  n = n > cap ? cap : n;
  if (slot0 >= n) return;

  // blurred 37-row x 40-byte windows (u in [-18, 21], v in [-18, 18]) of the group's keypoints, wavefront-private
  __shared__ uint32_t s_win[DK_WAVES][DK_PER_WAVE][DW_DWORDS];
  uint32_t(*win)[DW_DWORDS] = s_win[wave_in_block()];
  uvo_keypoint kp[DK_PER_WAVE];
  int64_t center_off[DK_PER_WAVE];
  int pitch[DK_PER_WAVE], ppitch[DK_PER_WAVE];
  const uint8_t* pbase[DK_PER_WAVE];
  float scale[DK_PER_WAVE];
  bool rescale[DK_PER_WAVE], live[DK_PER_WAVE];
  uint32_t px[DK_PER_WAVE][4];
  uint4 wv[DK_PER_WAVE][3];
  int wcy[DK_PER_WAVE], wcx[DK_PER_WAVE];  // (u, v) = (0, 0) inside the LDS window: row / byte column
  // tile rows of this lane: lane-load L = lane + 64 i covers row (L & 7) of tile (L >> 3) of the 4 x 6 grid, i.e. the 24 tiles are
  // spread over the three loads, eight lanes each
  int t_ty[3], t_tx[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = (lane + 64 * i) >> 3;
    t_ty[i] = t >> 2, t_tx[i] = t & 3;
  }
  // orientation-patch slots of this lane (same for every keypoint): row * 8 + chunk, see below
  int vv[4], u0[4];
  uint32_t pmask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slot = lane + 64 * i;
    vv[i] = (slot >> 3) - 15;
    u0[i] = -16 + 4 * (slot & 7);
    pmask[i] = patch[slot];
  }
  // ---- stage A: slot -> keypoint -> level geometry -> the patch / window loads.  Three dependent fetches per keypoint; written as
  // three passes over the group so that the four keypoints' fetches of a pass are in flight together (as one loop over the keypoints
  // the compiler chains them: every scalar load drains the previous keypoint's) ----
  FinalSlot fs[DK_PER_WAVE];
  if constexpr (DIRECT) {
    int lvl[DK_PER_WAVE], idx[DK_PER_WAVE];
#pragma unroll
    for (int k = 0; k < DK_PER_WAVE; ++k) {
      live[k] = slot0 + k < n;
      lvl[k] = 0, idx[k] = live[k] ? slot0 + k : slot0;  // dead entries repeat the first slot and are never stored
    }
#pragma unroll
    for (int l = 0; l + 1 < kMaxLevels; ++l)
#pragma unroll
      for (int k = 0; k < DK_PER_WAVE; ++k)
        if (lvl[k] == l && idx[k] >= lcount[l]) idx[k] -= lcount[l], lvl[k] = l + 1;
    uint32_t xy[DK_PER_WAVE], sc[DK_PER_WAVE];
#pragma unroll
    for (int k = 0; k < DK_PER_WAVE; ++k) {
      const int64_t o = (int64_t)f * ds.sel_block + lv[lvl[k]].sel_off + idx[k];
      xy[k] = ds.sel_xy[o], sc[k] = ds.sel_sc[o];
    }
#pragma unroll
    for (int k = 0; k < DK_PER_WAVE; ++k) {
      fs[k].x = (float)((int)(xy[k] & 0xffff) + kMinBorder), fs[k].y = (float)((int)(xy[k] >> 16) + kMinBorder);
      fs[k].level = lvl[k], fs[k].aux = (int32_t)sc[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < DK_PER_WAVE; ++k) {
      live[k] = slot0 + k < n;
      const int slot = live[k] ? slot0 + k : slot0;  // dead entries repeat the first slot and are never stored
      fs[k] = flist[(int64_t)f * flist_cap + slot];
    }
  }
  int level[DK_PER_WAVE];
  bool is_input[DK_PER_WAVE];
#pragma unroll
  for (int k = 0; k < DK_PER_WAVE; ++k) {
    is_input[k] = fs[k].level < 0;
    level[k] = __builtin_amdgcn_readfirstlane(is_input[k] ? 0 : fs[k].level);
    if (is_input[k]) {
      kp[k] = in_kp[(int64_t)f * in_cap + fs[k].aux];
    } else {
      kp[k].x = fs[k].x, kp[k].y = fs[k].y;
      kp[k].response = (float)fs[k].aux;
      kp[k].octave = level[k];
      kp[k].class_id = -1;
    }
    const LevelGeom& g = lv[level[k]];
    pitch[k] = g.pitch;
    scale[k] = g.scale;
    if (!is_input[k]) kp[k].size = g.patch_size;
    rescale[k] = !is_input[k] && level[k] != 0;
    center_off[k] = f * pyr_block + g.plane_off;  // completed below
    // level 0 read in place: the orientation patch (rows +-15, columns -16 .. +15 around a keypoint at least 16 pixels inside) lies in the
    // caller's image -- its own base and row pitch; the blurred window below always comes from the (tiled) blurred plane
    ppitch[k] = g.pitch, pbase[k] = pyr;
    if (level[k] == 0 && l0.vbase != nullptr) ppitch[k] = l0.pitch, pbase[k] = l0.vbase, center_off[k] = f * l0.frame_stride;
  }
#pragma unroll
  for (int k = 0; k < DK_PER_WAVE; ++k) {
    // (every lane holds the same keypoint; saying so puts the addresses below into scalar registers)
    const int cx = __builtin_amdgcn_readfirstlane(cv_round(kp[k].x)), cy = __builtin_amdgcn_readfirstlane(cv_round(kp[k].y));
    center_off[k] += (int64_t)(cy + kPad) * ppitch[k] + (cx + kPad);
    // Addresses = a wave-uniform base (18 rows above and 18 bytes left of the keypoint: the corner of the blurred window, which the
    // loads below reach anyway) + a non-negative 32-bit lane offset: one multiply-add per load instead of 64-bit arithmetic per lane.
    const int64_t corner_off = center_off[k] - (int64_t)18 * ppitch[k] - 18;
    // IC_Angle: the circular patch (rows v in [-15,15], |u| <= umax[|v|]) is read as 31 rows x 8 dwords starting at u = -16;
    // a table masks the bytes outside the circle (slots 248..255 = "row 31" are masked out entirely) -- applied in stage B
    const uint8_t* pcorner = pbase[k] + corner_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vr = vv[i] <= 15 ? vv[i] : 15;  // keep the (masked) load of row 31 inside the plane
      __builtin_memcpy(&px[k][i], pcorner + (uint32_t)((vr + 18) * ppitch[k] + (u0[i] + 18)), 4);
    }
    // blurred window: whole tiles of the tiled plane.  Window corner in padded coordinates (x0, y0) = (cx + 16 - 18, cy + 16 - 18);
    // tile grid origin (x0 / 16, y0 / 8); tiles beyond the window's last column / row are clamped onto the last one (loaded twice,
    // stored twice with the same bytes)
    const int x0 = cx + kPad - 18, y0 = cy + kPad - 18;
    const int tx0 = x0 >> 4, ty0 = y0 >> 3, ntx = ((x0 + 39) >> 4) - tx0, nty = ((y0 + 36) >> 3) - ty0;  // last tile index of the window in the grid
    wcx[k] = (x0 & 15) + 18, wcy[k] = (y0 & 7) + 18;
    const uint8_t* bplane = blur + f * pyr_block + lv[level[k]].plane_off;
    const int tiles_x = pitch[k] >> 4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ty = ty0 + (t_ty[i] < nty ? t_ty[i] : nty), tx = tx0 + (t_tx[i] < ntx ? t_tx[i] : ntx);
      wv[k][i] = *reinterpret_cast<const uint4*>(bplane + ((int64_t)ty * tiles_x + tx) * 128 + ((lane & 7) << 4));
    }
  }
#pragma unroll
  for (int k = 0; k < DK_PER_WAVE; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i)  // tile (t_ty, t_tx), row lane & 7 -> LDS row t_ty * 8 + (lane & 7), bytes t_tx * 16 ..
      *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(win[k]) + (t_ty[i] * 8 + (lane & 7)) * DW_PITCH + t_tx[i] * 16) = wv[k][i];
  // pattern: lane l evaluates pairs l, l+64, l+128, l+192
  float4 pq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) pq[j] = reinterpret_cast<const float4*>(pattern)[j * 64 + lane];  // (x0, y0, x1, y1)

  // ---- stage B: moments -> angle -> steering.  Per dword sum(I) and sum(k*I) by v_dot4_u32_u8, then
  // m10 += u0*sum(I) + sum(k*I), m01 += v*sum(I): exact int32 moments.  The 8 partial sums of the group (m10, m01 of 4
  // keypoints) are reduced together: every exchange step halves the number of values a lane carries (xor 32: 8 -> 4, xor 16:
  // 4 -> 2, xor 8: 2 -> 1, then xor 4/2/1 on the last one), 10 exchanges instead of 48, and value j ends up in lanes 8j..8j+7.
  // The angle and its sine / cosine are then computed once, on the lanes that own the keypoint, and broadcast. ----
  static_assert(DK_PER_WAVE == 4, "the grouped reduction and the descriptor store below are written for 4 keypoints");
  int mv[8];
#pragma unroll
  for (int k = 0; k < DK_PER_WAVE; ++k) {
    int m01 = 0, m10 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t pm = px[k][i] & pmask[i];  // the bytes of this dword inside the circle
      const int sum = (int)__builtin_amdgcn_udot4(pm, 0x01010101u, 0u, false);
      const int wsum = (int)__builtin_amdgcn_udot4(pm, 0x03020100u, 0u, false);
      m10 += u0[i] * sum + wsum;
      m01 += vv[i] * sum;
    }
    mv[2 * k] = m10, mv[2 * k + 1] = m01;
  }
  int r4[4], r2[2], r1;
  {
    const bool hi = (lane & 32) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int keep = hi ? mv[4 + j] : mv[j], send = hi ? mv[j] : mv[4 + j];
      r4[j] = keep + __shfl_xor(send, 32, 64);
    }
  }
  {
    const bool hi = (lane & 16) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int keep = hi ? r4[2 + j] : r4[j], send = hi ? r4[j] : r4[2 + j];
      r2[j] = keep + __shfl_xor(send, 16, 64);
    }
  }
  {
    const bool hi = (lane & 8) != 0;
    const int keep = hi ? r2[1] : r2[0], send = hi ? r2[0] : r2[1];
    r1 = keep + __shfl_xor(send, 8, 64);
  }
  r1 += __shfl_xor(r1, 4, 64);
  r1 += __shfl_xor(r1, 2, 64);
  r1 += __shfl_xor(r1, 1, 64);
  // lanes 16k..16k+7 hold m10 of keypoint k, lanes 16k+8..16k+15 its m01
  const int other = __shfl_xor(r1, 8, 64);
  const float my_m10 = (float)((lane & 8) ? other : r1), my_m01 = (float)((lane & 8) ? r1 : other);
  const float my_angle = uvo_fast_atan2(my_m01, my_m10);
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  float my_sin, my_cos;
  uvo_sincosf(my_angle * factorPI, &my_sin, &my_cos);
  float ca[DK_PER_WAVE], sa[DK_PER_WAVE];
#pragma unroll
  for (int k = 0; k < DK_PER_WAVE; ++k) {
    kp[k].angle = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_angle), 16 * k));
    ca[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_cos), 16 * k));  // a = cos
    sa[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_sin), 16 * k));  // b = sin
  }
  // ---- stage C: steered rBRIEF on the blurred level, sample points read from the LDS windows ----
  // cvRound (round half to even) and the LDS address in five fast-class instructions per sample point instead of six slow ones
  // (v_rndne + v_cvt_i32 twice, a shift and a three-operand add): adding 1.5 * 2^23 rounds a float of magnitude < 2^22 to an integer
  // under the default round-to-nearest-even mode -- ties go to the even integer because the constant is even -- and leaves that integer
  // in the low mantissa bits.  The low 16 bits of the two results are all the address needs: row * 64 + column + the byte address
  // of (u, v) = (0, 0) in the workgroup's LDS, in 16-bit arithmetic (negative rows / columns wrap and the sum comes out right).
  uint8_t t0[DK_PER_WAVE][4], t1[DK_PER_WAVE][4];
  const uint8_t* lds0 = reinterpret_cast<const uint8_t*>(&s_win[0][0][0]);
  const uint32_t wave_base = (uint32_t)(wave_in_block() * DK_PER_WAVE * DW_DWORDS * 4);
  float magic = 12582912.f;
  asm volatile("" : "+v"(magic));  // in a vector register: a fast-class instruction that reads a scalar register or a literal is a slow one
#pragma unroll
  for (int k = 0; k < DK_PER_WAVE; ++k) {
    uint32_t origin = wave_base + (uint32_t)(k * DW_DWORDS * 4 + wcy[k] * DW_PITCH + wcx[k]);
    asm volatile("" : "+v"(origin));
    const float a = ca[k], b = sa[k];
    auto sample = [&](float x, float y) -> uint8_t {
      const float fy = (x * b + y * a) + magic, fx = (x * a - y * b) + magic;
      uint32_t ad;
      asm("v_mul_lo_u16_e32 %0, 64, %1\n\tv_add_u16_e32 %0, %0, %2\n\tv_add_u16_e32 %0, %0, %3"
          : "=&v"(ad)
          : "v"(__builtin_bit_cast(uint32_t, fy)), "v"(__builtin_bit_cast(uint32_t, fx)), "v"(origin));
      return lds0[ad];
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t0[k][j] = sample(pq[j].x, pq[j].y);
      t1[k][j] = sample(pq[j].z, pq[j].w);
    }
  }
  // the group's descriptors are 128 consecutive bytes: the sixteen 64-bit ballots are dropped into lanes 0..15 (v_writelane takes the
  // scalar register the compare wrote) and leave in one store.  (s_nop 1: on gfx90a and later a vector instruction that reads a scalar
  // register must keep two wait states from the vector instruction that wrote it; the compiler inserts them for its own code only.)
  uint32_t dlo = 0, dhi = 0;
#define UVO_DESC_WORD(K, J)                                                                                         \
  {                                                                                                                 \
    const uint64_t w = ballot_mask(t0[K][J] < t1[K][J]);                                                            \
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"                                  \
        : "+v"(dlo), "+v"(dhi)                                                                                     \
        : "s"((uint32_t)w), "s"((uint32_t)(w >> 32)), "n"(4 * K + J));                                             \
  }
  UVO_DESC_WORD(0, 0) UVO_DESC_WORD(0, 1) UVO_DESC_WORD(0, 2) UVO_DESC_WORD(0, 3)
  UVO_DESC_WORD(1, 0) UVO_DESC_WORD(1, 1) UVO_DESC_WORD(1, 2) UVO_DESC_WORD(1, 3)
  UVO_DESC_WORD(2, 0) UVO_DESC_WORD(2, 1) UVO_DESC_WORD(2, 2) UVO_DESC_WORD(2, 3)
  UVO_DESC_WORD(3, 0) UVO_DESC_WORD(3, 1) UVO_DESC_WORD(3, 2) UVO_DESC_WORD(3, 3)
#undef UVO_DESC_WORD
  {
    const int nlive = n - slot0 < DK_PER_WAVE ? n - slot0 : DK_PER_WAVE;
    uint2* dd = reinterpret_cast<uint2*>(out_desc + ((int64_t)f * cap + slot0) * 32);
    if (lane < 4 * nlive) dd[lane] = make_uint2(dlo, dhi);
  }
#pragma unroll
  for (int k = 0; k < DK_PER_WAVE; ++k) {
    if (!live[k]) continue;
    const int slot = slot0 + k;
    if (lane == 0) {
      uvo_keypoint o = kp[k];
      if (rescale[k]) {
        o.x = o.x * scale[k];
        o.y = o.y * scale[k];
      }
      out_kp[(int64_t)f * cap + slot] = o;
    }
  }
}

// The occupancy grid the caller of the top-up extraction builds from its tracked keypoints (src/Tracking.cc:896-912): for every
// keypoint grid_2d((int)(pt.y / min_px_dist), (int)(pt.x / min_px_dist))++ -- Eigen::MatrixXi, column-major, one grid per frame.
// One workgroup per frame: the grid is cleared and filled in the same launch (Eigen::MatrixXi::Zero + the loop of :896-912).
__global__ __launch_bounds__(1024) void k_occupancy_grid(const uvo_keypoint* __restrict__ in_kp, const int32_t* __restrict__ n_in, int in_cap, int min_px_dist,
                                                         int grid_rows, int grid_cols, int32_t* __restrict__ grid) {
  const int f = blockIdx.x;
  int32_t* G = grid + (int64_t)f * grid_rows * grid_cols;
  for (int i = threadIdx.x; i < grid_rows * grid_cols; i += blockDim.x) G[i] = 0;
  __syncthreads();
  const int n = n_in ? min(n_in[f], in_cap) : 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const uvo_keypoint k = in_kp[(int64_t)f * in_cap + i];
    const int x = (int)(k.y / (float)min_px_dist), y = (int)(k.x / (float)min_px_dist);
    if (x >= 0 && x < grid_rows && y >= 0 && y < grid_cols) atomicAdd(&G[(int64_t)y * grid_rows + x], 1);
  }
}
void launch_occupancy_grid(hipStream_t s, const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int in_cap, int min_px_dist, int grid_rows, int grid_cols,
                           int32_t* d_grid, int batch) {
  hipLaunchKernelGGL(k_occupancy_grid, dim3(batch), dim3(1024), 0, s, d_in_kp, in_cap > 0 ? d_n_in : nullptr, in_cap, min_px_dist, grid_rows, grid_cols, d_grid);
}

void launch_assemble(hipStream_t s, const LevelGeom* d_lv, const Geom& g, FastAdapt fa, const uint32_t* d_sel_xy, const uint32_t* d_sel_sc,
                     const int32_t* d_sel_count, const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int in_cap, int32_t* d_grid, int grid_rows,
                     int grid_cols, int min_px_dist, int full_detect, const int32_t* d_nfn, FinalSlot* d_flist, int32_t* d_n_final,
                     int batch) {
  (void)d_in_kp;
  if (full_detect)
    hipLaunchKernelGGL(k_assemble<false>, dim3(batch), dim3(256), 0, s, d_lv, g.nlevels, fa, d_sel_xy, d_sel_sc, g.sel_block, d_sel_count, d_n_in, in_cap,
                       d_grid, grid_rows, grid_cols, min_px_dist, full_detect, d_nfn, d_flist, g.flist_cap, d_n_final);
  else
    hipLaunchKernelGGL(k_assemble<true>, dim3(batch), dim3(256), 0, s, d_lv, g.nlevels, fa, d_sel_xy, d_sel_sc, g.sel_block, d_sel_count, d_n_in, in_cap,
                       d_grid, grid_rows, grid_cols, min_px_dist, full_detect, d_nfn, d_flist, g.flist_cap, d_n_final);
}

void launch_describe(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint8_t* d_pyr, const uint8_t* d_blur, int64_t pyr_block,
                     const FinalSlot* d_flist, const int32_t* d_n_final, const uvo_keypoint* d_in_kp, int in_cap, const float* d_pattern,
                     const uint32_t* d_patch, uvo_keypoint* d_out_kp, uint8_t* d_out_desc, int cap, int32_t* d_n_out, int batch, Level0View l0) {
  const int slots = g.flist_cap < cap ? g.flist_cap : cap;
  hipLaunchKernelGGL(k_describe<false>, dim3((slots + DK_WAVES * DK_PER_WAVE - 1) / (DK_WAVES * DK_PER_WAVE), batch), dim3(64 * DK_WAVES), 0, s, d_lv, g.nlevels, d_pyr, d_blur, pyr_block, d_flist,
                     g.flist_cap, d_n_final, d_in_kp, in_cap, d_pattern, d_patch, d_out_kp, d_out_desc, cap, d_n_out, l0, DirectSel{});
}

// FullDetect without k_assemble (a frame or two): k_describe reads the quad-tree survivors itself
void launch_describe_direct(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint8_t* d_pyr, const uint8_t* d_blur, int64_t pyr_block, const uint32_t* d_sel_xy,
                            const uint32_t* d_sel_sc, const int32_t* d_sel_count, FastAdapt fa, const float* d_pattern, const uint32_t* d_patch,
                            uvo_keypoint* d_out_kp, uint8_t* d_out_desc, int cap, int32_t* d_n_out, int batch, Level0View l0) {
  const int slots = g.flist_cap < cap ? g.flist_cap : cap;
  const DirectSel ds{d_sel_xy, d_sel_sc, d_sel_count, g.sel_block, fa, batch};
  hipLaunchKernelGGL(k_describe<true>, dim3((slots + DK_WAVES * DK_PER_WAVE - 1) / (DK_WAVES * DK_PER_WAVE), batch), dim3(64 * DK_WAVES), 0, s, d_lv, g.nlevels, d_pyr, d_blur, pyr_block,
                     (const FinalSlot*)nullptr, g.flist_cap, (const int32_t*)nullptr, (const uvo_keypoint*)nullptr, 0, d_pattern, d_patch, d_out_kp, d_out_desc, cap, d_n_out, l0, ds);
}

}  // namespace uvo
