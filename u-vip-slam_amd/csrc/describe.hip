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
__global__ __launch_bounds__(256) void k_assemble(const LevelGeom* __restrict__ lv, int nlevels, const uint32_t* __restrict__ sel_xy,
                                                  const uint32_t* __restrict__ sel_sc, int sel_block,
                                                  const int32_t* __restrict__ sel_count, const int32_t* __restrict__ n_in, int in_cap,
                                                  int32_t* __restrict__ grid, int grid_rows, int grid_cols, int min_px_dist,
                                                  int full_detect, const int32_t* __restrict__ nfn, FinalSlot* __restrict__ flist,
                                                  int flist_cap, int32_t* __restrict__ n_final) {
  const int f = blockIdx.x;
  const uint32_t* sxy = sel_xy + (int64_t)f * sel_block;
  const uint32_t* ssc = sel_sc + (int64_t)f * sel_block;
  const int32_t* cnt = sel_count + f * nlevels;
  FinalSlot* out = flist + (int64_t)f * flist_cap;
  if (full_detect) {
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
  // top-up mode
  const int nin = n_in ? min(n_in[f], in_cap) : 0;
  for (int i = threadIdx.x; i < nin; i += blockDim.x) {
    FinalSlot s;
    s.x = 0.f, s.y = 0.f;  // coordinates are read from in_kp by k_describe
    s.level = (int32_t)0x80000000;
    s.aux = i;
    if (i < flist_cap) out[i] = s;
  }
  if (threadIdx.x != 0) return;
  int32_t* G = grid + (int64_t)f * grid_rows * grid_cols;
  const int need = nfn[f];
  int pos = nin;
  int Total_counter = 0, KP_counter = 0;
  bool break_key = false;
  for (int l = 0; l < nlevels && !break_key; ++l) {
    int n = cnt[l];
    n = n > lv[l].sel_cap ? lv[l].sel_cap : n;
    if (n == 0) continue;
    const int numofpoint = need * (8 - l) / 30;
    const float scale = lv[l].scale;
    for (int i = 0; i < n; ++i) {
      const uint32_t xy = sxy[lv[l].sel_off + i];
      const float px = (float)((int)(xy & 0xffff) + kMinBorder), py = (float)((int)(xy >> 16) + kMinBorder);
      const float tx = px * scale, ty = py * scale;
      const int r = (int)(ty / (float)min_px_dist), c = (int)(tx / (float)min_px_dist);
      int32_t* cell = &G[(int64_t)c * grid_rows + r];  // Eigen column-major
      if (*cell > 0) continue;
      FinalSlot s;
      s.x = px, s.y = py, s.level = l, s.aux = (int32_t)ssc[lv[l].sel_off + i];
      if (pos < flist_cap) out[pos] = s;
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

// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// 4 wavefronts per workgroup, one final slot per wavefront.
__global__ __launch_bounds__(256) void k_describe(const LevelGeom* __restrict__ lv, int nlevels, const uint8_t* __restrict__ pyr,
                                                  const uint8_t* __restrict__ blur, int64_t pyr_block,
                                                  const FinalSlot* __restrict__ flist, int flist_cap, const int32_t* __restrict__ n_final,
                                                  const uvo_keypoint* __restrict__ in_kp, int in_cap, const int8_t* __restrict__ pattern,
                                                  const uint16_t* __restrict__ patch, uvo_keypoint* __restrict__ out_kp,
                                                  uint8_t* __restrict__ out_desc, int cap, int32_t* __restrict__ n_out) {
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + wave_in_block();
  int n = n_final[f];
  if (blockIdx.x == 0 && threadIdx.x == 0) n_out[f] = n;
  n = n > flist_cap ? flist_cap : n;
  if (slot >= n || slot >= cap) return;
  const FinalSlot fs = flist[(int64_t)f * flist_cap + slot];
  const bool is_input = fs.level < 0;
  uvo_keypoint kp;
  int level;
  if (is_input) {
    kp = in_kp[(int64_t)f * in_cap + fs.aux];
    level = 0;
  } else {
    level = fs.level;
    kp.x = fs.x, kp.y = fs.y;
    kp.size = lv[level].patch_size;
    kp.response = (float)fs.aux;
    kp.octave = level;
    kp.class_id = -1;
  }
  const LevelGeom& g = lv[level];
  const int cx = cv_round(kp.x), cy = cv_round(kp.y);
  const int64_t center_off = f * pyr_block + g.plane_off + (int64_t)(cy + kPad) * g.pitch + (cx + kPad);

  // ---- IC_Angle: the 749 (u, v) offsets of the circular patch (rows v in [-15,15], |u| <= umax[|v|]) come from a
  // table padded to 768 entries with (0,0) (contributes nothing); 12 independent byte loads per lane ----
  {
    const uint8_t* center = pyr + center_off;
    int m01 = 0, m10 = 0;
    int pix[12], uu[12], vv[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int uv = patch[lane + 64 * i];  // u in the low byte, v in the high byte (both int8)
      uu[i] = (int)(int8_t)(uv & 0xff);
      vv[i] = (int)(int8_t)(uv >> 8);
      pix[i] = center[(int64_t)vv[i] * g.pitch + uu[i]];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      m10 += uu[i] * pix[i];
      m01 += vv[i] * pix[i];
    }
    m01 = wave_sum(m01);
    m10 = wave_sum(m10);
    kp.angle = uvo_fast_atan2((float)m01, (float)m10);
  }

  // ---- steered rBRIEF on the blurred level: lane l evaluates pairs l, l+64, l+128, l+192 ----
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float angle = kp.angle * factorPI;
  float a, b;
  uvo_sincosf(angle, &b, &a);  // a = cos, b = sin
  const uint8_t* bc = blur + center_off;
  uint64_t words[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t pq = reinterpret_cast<const uint32_t*>(pattern)[j * 64 + lane];  // (x0, y0, x1, y1) int8
    const float x0 = (float)(int8_t)(pq & 0xff), y0 = (float)(int8_t)((pq >> 8) & 0xff), x1 = (float)(int8_t)((pq >> 16) & 0xff),
                y1 = (float)(int8_t)(pq >> 24);
    const int t0 = bc[(int64_t)cv_round(x0 * b + y0 * a) * g.pitch + cv_round(x0 * a - y0 * b)];
    const int t1 = bc[(int64_t)cv_round(x1 * b + y1 * a) * g.pitch + cv_round(x1 * a - y1 * b)];
    words[j] = __ballot(t0 < t1);
  }
  uint64_t* dd = reinterpret_cast<uint64_t*>(out_desc + ((int64_t)f * cap + slot) * 32);
  if (lane < 4) dd[lane] = words[lane];

  if (lane == 0) {
    if (!is_input && level != 0) {
      kp.x = kp.x * g.scale;
      kp.y = kp.y * g.scale;
    }
    out_kp[(int64_t)f * cap + slot] = kp;
  }
}

void launch_assemble(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint32_t* d_sel_xy, const uint32_t* d_sel_sc,
                     const int32_t* d_sel_count, const uvo_keypoint* d_in_kp, const int32_t* d_n_in, int in_cap, int32_t* d_grid, int grid_rows,
                     int grid_cols, int min_px_dist, int full_detect, const int32_t* d_nfn, FinalSlot* d_flist, int32_t* d_n_final,
                     int batch) {
  (void)d_in_kp;
  hipLaunchKernelGGL(k_assemble, dim3(batch), dim3(256), 0, s, d_lv, g.nlevels, d_sel_xy, d_sel_sc, g.sel_block, d_sel_count, d_n_in, in_cap,
                     d_grid, grid_rows, grid_cols, min_px_dist, full_detect, d_nfn, d_flist, g.flist_cap, d_n_final);
}

void launch_describe(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint8_t* d_pyr, const uint8_t* d_blur, int64_t pyr_block,
                     const FinalSlot* d_flist, const int32_t* d_n_final, const uvo_keypoint* d_in_kp, int in_cap, const int8_t* d_pattern,
                     const uint16_t* d_patch, uvo_keypoint* d_out_kp, uint8_t* d_out_desc, int cap, int32_t* d_n_out, int batch) {
  const int slots = g.flist_cap < cap ? g.flist_cap : cap;
  hipLaunchKernelGGL(k_describe, dim3((slots + 3) / 4, batch), dim3(256), 0, s, d_lv, g.nlevels, d_pyr, d_blur, pyr_block, d_flist,
                     g.flist_cap, d_n_final, d_in_kp, in_cap, d_pattern, d_patch, d_out_kp, d_out_desc, cap, d_n_out);
}

}  // namespace uvo
