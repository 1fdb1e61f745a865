// cv::CLAHE::apply for 8-bit images (OpenCV 3.4 imgproc/src/clahe.cpp), the pre-processing step of Tracking::GrabImage when
// Enhance = 1 (src/Tracking.cc:425-431: clip limit 4, 12 x 12 tiles).
//   k_clahe_lut   : one workgroup per (tile, frame): 256-bin histogram of the tile (the image is extended to a multiple of the
//                   tile grid on the right / bottom by REFLECT_101, exactly as apply() does with copyMakeBorder), clip at
//                   clipLimit, redistribute the excess (batch + strided residual), running sum -> LUT byte = cvRound(sum * scale)
//   k_clahe_apply : per pixel the four neighbouring tile LUTs blended bilinearly in fp32, in the reference's expression order
#include "common.hpp"
#include "uvo_math.hpp"

namespace uvo {

__device__ __forceinline__ int reflect101r(int p, int len) {  // right / bottom extension only: p >= 0
  return p < len ? p : 2 * (len - 1) - p;
}

__global__ __launch_bounds__(256) void k_clahe_lut(const uint8_t* __restrict__ src, int w, int h, int64_t stride, int64_t frame_stride, int tiles_x,
                                                   int tile_w, int tile_h, int clip_limit, float lut_scale, uint8_t* __restrict__ lut,
                                                   int64_t lut_frame) {
  __shared__ int s_hist[256];
  __shared__ int s_part[4];
  const int k = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
  const int ty = k / tiles_x, tx = k - ty * tiles_x;
  s_hist[tid] = 0;
  __syncthreads();
  const uint8_t* img = src + f * frame_stride;
  const int n = tile_w * tile_h;
  for (int i = tid; i < n; i += 256) {
    const int yy = i / tile_w, xx = i - yy * tile_w;
    const int x = reflect101r(tx * tile_w + xx, w), y = reflect101r(ty * tile_h + yy, h);
    atomicAdd(&s_hist[img[(int64_t)y * stride + x]], 1);
  }
  __syncthreads();
  int hv = s_hist[tid];
  if (clip_limit > 0) {
    int excess = hv > clip_limit ? hv - clip_limit : 0;
    hv = hv > clip_limit ? clip_limit : hv;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) excess += __shfl_xor(excess, off, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = excess;
    __syncthreads();
    const int clipped = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int redistBatch = clipped / 256;
    int residual = clipped - redistBatch * 256;
    hv += redistBatch;
    if (residual != 0) {
      const int residualStep = max(256 / residual, 1);
      if (tid % residualStep == 0 && tid / residualStep < residual) hv++;  // for (i = 0; i < 256 && residual > 0; i += step, residual--)
    }
    __syncthreads();
  }
  // inclusive running sum over the 256 bins
  int incl = hv;
  const int lane = tid & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_part[tid >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < (tid >> 6); ++q) base += s_part[q];
  const int sum = base + incl;
  int r = cv_round((float)sum * lut_scale);  // saturate_cast<uchar>(sum * lutScale_)
  r = r < 0 ? 0 : (r > 255 ? 255 : r);
  lut[f * lut_frame + (int64_t)k * 256 + tid] = (uint8_t)r;
}

__global__ __launch_bounds__(256) void k_clahe_apply(const uint8_t* __restrict__ src, int w, int h, int64_t stride, int64_t frame_stride,
                                                     int tiles_x, int tiles_y, int tile_w, int tile_h, const uint8_t* __restrict__ lut,
                                                     int64_t lut_frame, uint8_t* __restrict__ dst, int64_t dst_stride,
                                                     int64_t dst_frame_stride) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_in_block(), f = blockIdx.z;
  if (x >= w || y >= h) return;
  const float inv_tw = 1.0f / (float)tile_w, inv_th = 1.0f / (float)tile_h;
  const float txf = (float)x * inv_tw - 0.5f;
  int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
  const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
  tx1 = max(tx1, 0), tx2 = min(tx2, tiles_x - 1);
  const float tyf = (float)y * inv_th - 0.5f;
  int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
  const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
  ty1 = max(ty1, 0), ty2 = min(ty2, tiles_y - 1);
  const int v = src[f * frame_stride + (int64_t)y * stride + x];
  const uint8_t* L = lut + f * lut_frame + v;
  const float p11 = (float)L[(int64_t)(ty1 * tiles_x + tx1) * 256], p12 = (float)L[(int64_t)(ty1 * tiles_x + tx2) * 256];
  const float p21 = (float)L[(int64_t)(ty2 * tiles_x + tx1) * 256], p22 = (float)L[(int64_t)(ty2 * tiles_x + tx2) * 256];
  const float res = (p11 * xa1 + p12 * xa) * ya1 + (p21 * xa1 + p22 * xa) * ya;
  int r = cv_round(res);
  r = r < 0 ? 0 : (r > 255 ? 255 : r);
  dst[f * dst_frame_stride + (int64_t)y * dst_stride + x] = (uint8_t)r;
}

void launch_clahe(hipStream_t s, const uint8_t* d_src, int w, int h, int64_t stride, int64_t frame_stride, int batch, int tiles_x, int tiles_y,
                  int tile_w, int tile_h, int clip_limit, float lut_scale, uint8_t* d_lut, uint8_t* d_dst, int64_t dst_stride,
                  int64_t dst_frame_stride) {
  const int64_t lut_frame = (int64_t)tiles_x * tiles_y * 256;
  hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y, batch), dim3(256), 0, s, d_src, w, h, stride, frame_stride, tiles_x, tile_w, tile_h,
                     clip_limit, lut_scale, d_lut, lut_frame);
  hipLaunchKernelGGL(k_clahe_apply, dim3((w + 63) / 64, (h + 3) / 4, batch), dim3(256), 0, s, d_src, w, h, stride, frame_stride, tiles_x, tiles_y,
                     tile_w, tile_h, d_lut, lut_frame, d_dst, dst_stride, dst_frame_stride);
}

}  // namespace uvo
