// The KLT step in front of the extractor:
//   cv::buildOpticalFlowPyramid(im, imgpyr, mWin_Size, mPyr_Levels)                                   src/FrameKTL.cc:76
//   cv::calcOpticalFlowPyrLK(img0pyr, img1pyr, pts0, pts1, mask_klt, error, win_size, pyr_levels,
//        TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW + OPTFLOW_LK_GET_MIN_EIGENVALS)   src/Tracking.cc:1046-1047
// (OpenCV 3.4 video/src/lkpyramid.cpp, imgproc/src/pyramids.cpp).
//   k_klt_level0 / k_klt_pyrdown / k_klt_border : image levels with a winSize REFLECT_101 border (pyrDown = 1-4-6-4-1 taps,
//       (sum + 128) >> 8; reading the bordered previous level gives pyrDown's own reflection for free)
//   k_klt_scharr : calcSharrDeriv, interleaved int16 (dx, dy), zero border
//   k_klt_track  : one wavefront per point, all pyramid levels and iterations inside the kernel.  The window pixels are spread
//       over the lanes (21 x 21 = 441 -> 7 per lane); the template patch I, dI is interpolated once per level in the reference's
//       14-bit fixed point and kept in registers, every iteration interpolates J the same way, and the sums A11, A12, A22, b1,
//       b2 are reduced per lane then across the wavefront.  All integer quantities are exact; the float sums are associated
//       differently from a raster-order CPU loop, so positions agree with the CPU statement to float rounding (tolerance in
//       the parity test), as they do between OpenCV's own scalar and SIMD builds.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "uvo_math.hpp"

namespace uvo {

struct KltLevel {
  int w, h;
  int ipitch;       // image bytes per row (incl. border)
  int dpitch;       // derivative shorts per row (incl. border)
  int64_t ioff;     // byte offset of the level's bordered image inside the slot's image block
  int64_t doff;     // short offset of the level's bordered derivative plane inside the slot's derivative block
};
constexpr int kKltMaxLevels = 8;
struct KltGeom {
  KltLevel l[kKltMaxLevels];
  int nlevels, bx, by;
};

__device__ __forceinline__ int klt_reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// level 0: caller image -> bordered buffer (interior + REFLECT_101 border), one thread per byte of the bordered plane
__global__ __launch_bounds__(256) void k_klt_level0(const uint8_t* __restrict__ img, int w, int h, int64_t stride, uint8_t* __restrict__ dst, int pitch,
                                                    int bx, int by) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w + 2 * bx) return;
  dst[(int64_t)y * pitch + x] = img[(int64_t)klt_reflect101(y - by, h) * stride + klt_reflect101(x - bx, w)];
}

// pyrDown of the bordered previous level into the interior of this one
__global__ __launch_bounds__(256) void k_klt_pyrdown(const uint8_t* __restrict__ src, int spitch, uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                                     int bx, int by) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= dw) return;
  const uint8_t* s = src + (int64_t)(by + 2 * y - 2) * spitch + bx + 2 * x - 2;
  int rows[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const uint8_t* r = s + (int64_t)k * spitch;
    rows[k] = r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4];
  }
  dst[(int64_t)(by + y) * dpitch + bx + x] = (uint8_t)((rows[2] * 6 + (rows[1] + rows[3]) * 4 + rows[0] + rows[4] + 128) >> 8);
}

// REFLECT_101 border of a level from its own interior (copyMakeBorder(..., pyrBorder | BORDER_ISOLATED))
__global__ __launch_bounds__(256) void k_klt_border(uint8_t* __restrict__ lvl, int w, int h, int pitch, int bx, int by) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w + 2 * bx) return;
  const int sx = klt_reflect101(x - bx, w), sy = klt_reflect101(y - by, h);
  if (sx == x - bx && sy == y - by) return;  // interior
  lvl[(int64_t)y * pitch + x] = lvl[(int64_t)(sy + by) * pitch + sx + bx];
}

// calcSharrDeriv on the level (its REFLECT_101 border supplies the edge rule), interior of the derivative plane
__global__ __launch_bounds__(256) void k_klt_scharr(const uint8_t* __restrict__ lvl, int w, int h, int pitch, int bx, int by,
                                                    int16_t* __restrict__ deriv, int dpitch) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const uint8_t* p = lvl + (int64_t)(y + by) * pitch + x + bx;
  const uint8_t *r0 = p - pitch, *r2 = p + pitch;
  const int t0m = (r0[-1] + r2[-1]) * 3 + p[-1] * 10, t0p = (r0[1] + r2[1]) * 3 + p[1] * 10;
  const int t1m = r2[-1] - r0[-1], t1c = r2[0] - r0[0], t1p = r2[1] - r0[1];
  int16_t* d = deriv + (int64_t)(y + by) * dpitch + 2 * (x + bx);
  d[0] = (int16_t)(t0p - t0m);
  d[1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

#define KLT_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// NP = window pixels per lane (win_w * win_h <= 64 * NP)
template <int NP>
__global__ __launch_bounds__(256) void k_klt_track(KltGeom G, const uint8_t* __restrict__ img0, const int16_t* __restrict__ der0,
                                                   const uint8_t* __restrict__ img1, const float* __restrict__ prev_pts,
                                                   float* __restrict__ next_pts, int npts, int win_w, int win_h, int max_level, int max_count,
                                                   float epsilon_sq, float min_eig_thr, uint8_t* __restrict__ status, float* __restrict__ err) {
  const int pt = blockIdx.x * 4 + wave_in_block(), lane = threadIdx.x & 63;
  if (pt >= npts) return;
  const int npx = win_w * win_h;
  int wx[NP], wy[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int idx = lane + 64 * k;
    wy[k] = idx / win_w;
    wx[k] = idx - wy[k] * win_w;
  }
  const float halfx = (float)(win_w - 1) * 0.5f, halfy = (float)(win_h - 1) * 0.5f;
  const float ppx = prev_pts[2 * pt], ppy = prev_pts[2 * pt + 1];
  float nx = next_pts[2 * pt], ny = next_pts[2 * pt + 1];  // OPTFLOW_USE_INITIAL_FLOW
  bool ok = true;
  float e_out = 0.f;
  const int W_BITS = 14, W_BITS1 = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  for (int level = max_level; level >= 0; --level) {
    const KltLevel L = G.l[level];
    const float sc = (float)(1. / (double)(1 << level));
    float prevx = ppx * sc, prevy = ppy * sc;
    float nextx, nexty;
    if (level == max_level) {
      nextx = nx * sc, nexty = ny * sc;
    } else {
      nextx = nx * 2.f, nexty = ny * 2.f;
    }
    nx = nextx, ny = nexty;
    prevx -= halfx, prevy -= halfy;
    const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
    if (ipx < -win_w || ipx >= L.w || ipy < -win_h || ipy >= L.h) {
      if (level == 0) ok = false, e_out = 0.f;
      continue;
    }
    float a = prevx - (float)ipx, b = prevy - (float)ipy;
    int iw00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    int iw01 = cv_round(a * (1.f - b) * (float)(1 << W_BITS));
    int iw10 = cv_round((1.f - a) * b * (float)(1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    const uint8_t* Ibase = img0 + L.ioff + (int64_t)(G.by + ipy) * L.ipitch + G.bx + ipx;
    const int16_t* Dbase = der0 + L.doff + (int64_t)(G.by + ipy) * L.dpitch + 2 * (G.bx + ipx);
    int Iv[NP], Ix[NP], Iy[NP];
    float iA11 = 0.f, iA12 = 0.f, iA22 = 0.f;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      Iv[k] = 0, Ix[k] = 0, Iy[k] = 0;
      if (lane + 64 * k < npx) {
        const uint8_t* s = Ibase + (int64_t)wy[k] * L.ipitch + wx[k];
        const int16_t* d = Dbase + (int64_t)wy[k] * L.dpitch + 2 * wx[k];
        const int ival = KLT_DESCALE(s[0] * iw00 + s[1] * iw01 + s[L.ipitch] * iw10 + s[L.ipitch + 1] * iw11, W_BITS1 - 5);
        const int ixval = KLT_DESCALE(d[0] * iw00 + d[2] * iw01 + d[L.dpitch] * iw10 + d[L.dpitch + 2] * iw11, W_BITS1);
        const int iyval = KLT_DESCALE(d[1] * iw00 + d[3] * iw01 + d[L.dpitch + 1] * iw10 + d[L.dpitch + 3] * iw11, W_BITS1);
        Iv[k] = (int)(int16_t)ival, Ix[k] = (int)(int16_t)ixval, Iy[k] = (int)(int16_t)iyval;
        iA11 += (float)(Ix[k] * Ix[k]);
        iA12 += (float)(Ix[k] * Iy[k]);
        iA22 += (float)(Iy[k] * Iy[k]);
      }
    }
    const float A11 = wave_sum_f(iA11) * FLT_SCALE, A12 = wave_sum_f(iA12) * FLT_SCALE, A22 = wave_sum_f(iA22) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win_w * win_h);
    e_out = minEig;  // OPTFLOW_LK_GET_MIN_EIGENVALS
    if (minEig < min_eig_thr || D < 1.1920929e-07f) {
      if (level == 0) ok = false;
      continue;
    }
    D = 1.f / D;
    nextx -= halfx, nexty -= halfy;
    float pdx = 0.f, pdy = 0.f;
    const uint8_t* Jlvl = img1 + L.ioff;
    for (int j = 0; j < max_count; ++j) {
      const int inx = (int)floorf(nextx), iny = (int)floorf(nexty);
      if (inx < -win_w || inx >= L.w || iny < -win_h || iny >= L.h) {
        if (level == 0) ok = false;
        break;
      }
      a = nextx - (float)inx, b = nexty - (float)iny;
      iw00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
      iw01 = cv_round(a * (1.f - b) * (float)(1 << W_BITS));
      iw10 = cv_round((1.f - a) * b * (float)(1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      const uint8_t* Jbase = Jlvl + (int64_t)(G.by + iny) * L.ipitch + G.bx + inx;
      float ib1 = 0.f, ib2 = 0.f;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (lane + 64 * k < npx) {
          const uint8_t* s = Jbase + (int64_t)wy[k] * L.ipitch + wx[k];
          const int diff = KLT_DESCALE(s[0] * iw00 + s[1] * iw01 + s[L.ipitch] * iw10 + s[L.ipitch + 1] * iw11, W_BITS1 - 5) - Iv[k];
          ib1 += (float)(diff * Ix[k]);
          ib2 += (float)(diff * Iy[k]);
        }
      }
      const float b1 = wave_sum_f(ib1) * FLT_SCALE, b2 = wave_sum_f(ib2) * FLT_SCALE;
      const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
      nextx += dx, nexty += dy;
      nx = nextx + halfx, ny = nexty + halfy;
      if ((double)dx * (double)dx + (double)dy * (double)dy <= (double)epsilon_sq) break;
      if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
        nx -= dx * 0.5f, ny -= dy * 0.5f;
        break;
      }
      pdx = dx, pdy = dy;
    }
  }
  if (lane == 0) {
    next_pts[2 * pt] = nx, next_pts[2 * pt + 1] = ny;
    status[pt] = ok ? 1 : 0;
    err[pt] = e_out;
  }
}

// Tracking::undistort_point (src/Tracking.cc:1265-1283) for n points: cv::undistortPoints(pt, pt, K, D, noArray(), K) or, for the
// fisheye model, cv::fisheye::undistortPoints(pt, pt, K, D, Mat(), K) -- one thread per point, double arithmetic like OpenCV 3.4's
// (cvUndistortPointsInternal: five fixed-point iterations of the Brown model, no epsilon test; fisheye: at most ten Newton steps on
// theta, |step| < 1e-8 ends them, theta_d clamped to [-pi/2, pi/2]), re-projected with P = K.  [OCV-RECALL: unpinned, see tools/pin]
struct UndistortCam {
  double fx, fy, cx, cy, k[8];
  int fisheye;
};
__global__ __launch_bounds__(256) void k_undistort(UndistortCam C, const float* __restrict__ pts, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double u = (double)pts[2 * i], v = (double)pts[2 * i + 1];
  double x, y;
  if (!C.fisheye) {
    const double ifx = 1. / C.fx, ify = 1. / C.fy;
    x = (u - C.cx) * ifx, y = (v - C.cy) * ify;
    const double x0 = x, y0 = y;
    const double* k = C.k;  // k1 k2 p1 p2 k3 k4 k5 k6
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = C.fx * x + 0. * y + C.cx, yy = 0. * x + C.fy * y + C.cy, ww = 1. / (0. * x + 0. * y + 1.);
    out[2 * i] = (float)(xx * ww), out[2 * i + 1] = (float)(yy * ww);
  } else {
    const double pwx = (u - C.cx) / C.fx, pwy = (v - C.cy) / C.fy;
    double scale = 1.0;
    double theta_d = sqrt(pwx * pwx + pwy * pwy);
    const double half_pi = 3.1415926535897932384626433832795 / 2.;
    theta_d = fmin(fmax(-half_pi, theta_d), half_pi);
    if (theta_d > 1e-8) {
      double theta = theta_d;
      for (int j = 0; j < 10; ++j) {
        const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
        const double k0_theta2 = C.k[0] * theta2, k1_theta4 = C.k[1] * theta4, k2_theta6 = C.k[2] * theta6, k3_theta8 = C.k[3] * theta8;
        const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                 (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
        theta = theta - theta_fix;
        if (fabs(theta_fix) < 1e-8) break;
      }
      scale = tan(theta) / theta_d;
    }
    x = pwx * scale, y = pwy * scale;
    const double pr0 = C.fx * x + 0. * y + C.cx * 1.0, pr1 = 0. * x + C.fy * y + C.cy * 1.0, pr2 = 0. * x + 0. * y + 1. * 1.0;
    out[2 * i] = (float)(pr0 / pr2), out[2 * i + 1] = (float)(pr1 / pr2);
  }
}

}  // namespace uvo

// ---------------------------------------------------------------------------------------------------------------------------
using namespace uvo;

struct uvo_klt {
  uvo_klt_cfg cfg;
  hipStream_t stream = nullptr;
  KltGeom geom;          // for the maximum image size: offsets fixed at create
  int64_t img_block = 0, der_block = 0;  // per slot
  uint8_t* d_img = nullptr;    // [slots][img_block]
  int16_t* d_der = nullptr;    // [slots][der_block]
  uint8_t* d_in = nullptr;     // staging of the caller image
  // point arrays of one track call, one device block [prev | next | err | status] with a page-locked mirror: one copy each way
  uint8_t *d_pts = nullptr, *h_pts = nullptr;
  float *d_prev = nullptr, *d_next = nullptr, *d_err = nullptr;
  uint8_t* d_status = nullptr;
  std::vector<int> slot_w, slot_h, slot_levels;
};

static void klt_geometry(int w, int h, int bx, int by, int max_level, KltGeom& G, int64_t* img_bytes, int64_t* der_shorts) {
  G.bx = bx, G.by = by;
  int64_t io = 0, dof = 0;
  int lw = w, lh = h, n = 0;
  for (int l = 0; l <= max_level && l < kKltMaxLevels; ++l) {
    KltLevel& L = G.l[l];
    L.w = lw, L.h = lh;
    L.ipitch = (lw + 2 * bx + 63) / 64 * 64;
    L.dpitch = 2 * (lw + 2 * bx);
    L.ioff = io, L.doff = dof;
    io += (int64_t)L.ipitch * (lh + 2 * by) + 256;
    dof += (int64_t)L.dpitch * (lh + 2 * by) + 128;
    ++n;
    lw = (lw + 1) / 2, lh = (lh + 1) / 2;
    if (lw <= bx || lh <= by) break;  // buildOpticalFlowPyramid: stop when the next level would not exceed the window
  }
  G.nlevels = n;
  *img_bytes = io, *der_shorts = dof;
}

extern "C" {

void uvo_klt_destroy(uvo_klt* k) {
  if (!k) return;
  hipSetDevice(k->cfg.device);
  if (k->stream) hipStreamSynchronize(k->stream);
  void* ptrs[] = {k->d_img, k->d_der, k->d_in, k->d_pts};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (k->h_pts) (void)hipHostFree(k->h_pts);
  if (k->stream) hipStreamDestroy(k->stream);
  delete k;
}

int uvo_klt_create(const uvo_klt_cfg* cfg, uvo_klt** out) {
  if (!cfg || !out) return fail(UVO_E_BADARG, "null pointer");
  *out = nullptr;
  if (cfg->max_width < 8 || cfg->max_height < 8 || cfg->max_level < 0 || cfg->max_level >= kKltMaxLevels || cfg->win_width < 3 ||
      cfg->win_height < 3 || cfg->win_width * cfg->win_height > 1024 || cfg->max_points < 1 || cfg->slots < 2 || cfg->slots > 16)
    return fail(UVO_E_BADARG, "bad KLT configuration (window at most 1024 pixels, 2..16 pyramid slots)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(UVO_E_NODEVICE, "no HIP device available (no CPU fallback exists)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(UVO_E_BADARG, "device ordinal out of range");
  uvo_klt* k = new uvo_klt();
  k->cfg = *cfg;
  if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking) != hipSuccess) {
    delete k;
    return fail(UVO_E_HIP, "stream creation failed");
  }
  klt_geometry(cfg->max_width, cfg->max_height, cfg->win_width, cfg->win_height, cfg->max_level, k->geom, &k->img_block, &k->der_block);
  k->slot_w.assign(cfg->slots, 0), k->slot_h.assign(cfg->slots, 0), k->slot_levels.assign(cfg->slots, 0);
  const size_t S = (size_t)cfg->slots, N = (size_t)cfg->max_points;
  if (hipMalloc((void**)&k->d_img, S * k->img_block) != hipSuccess || hipMalloc((void**)&k->d_der, S * k->der_block * 2) != hipSuccess ||
      hipMalloc((void**)&k->d_in, (size_t)cfg->max_width * cfg->max_height) != hipSuccess || hipMalloc((void**)&k->d_pts, N * 37 + 64) != hipSuccess ||
      hipHostMalloc((void**)&k->h_pts, N * 37 + 64, hipHostMallocDefault) != hipSuccess) {
    uvo_klt_destroy(k);
    return fail(UVO_E_NOMEM, "KLT scratch allocation failed");
  }
  *out = k;
  return UVO_OK;
}

// d_src: device image (tight or pitched rows) already ordered before the handle's stream
static int klt_build_from_device(uvo_klt* k, int slot, const uint8_t* d_src, int width, int height, int64_t src_pitch, int* levels_built) {
  hipStream_t s = k->stream;
  KltGeom G;
  int64_t ib, db;
  klt_geometry(width, height, k->cfg.win_width, k->cfg.win_height, k->cfg.max_level, G, &ib, &db);
  if (ib > k->img_block || db > k->der_block) return fail(UVO_E_BADARG, "image size outside what the handle was sized for");
  uint8_t* I = k->d_img + (int64_t)slot * k->img_block;
  int16_t* D = k->d_der + (int64_t)slot * k->der_block;
  const int bx = G.bx, by = G.by;
  UVO_HIP_CHECK(hipMemsetAsync(D, 0, (size_t)db * 2, s));  // derivBorder = BORDER_CONSTANT (zeros)
  for (int l = 0; l < G.nlevels; ++l) {
    const KltLevel& L = G.l[l];
    const dim3 gb((L.w + 2 * bx + 255) / 256, L.h + 2 * by), gi((L.w + 255) / 256, L.h);
    if (l == 0) {
      hipLaunchKernelGGL(k_klt_level0, gb, dim3(256), 0, s, d_src, width, height, src_pitch, I + L.ioff, L.ipitch, bx, by);
    } else {
      const KltLevel& P = G.l[l - 1];
      hipLaunchKernelGGL(k_klt_pyrdown, gi, dim3(256), 0, s, I + P.ioff, P.ipitch, I + L.ioff, L.w, L.h, L.ipitch, bx, by);
      hipLaunchKernelGGL(k_klt_border, gb, dim3(256), 0, s, I + L.ioff, L.w, L.h, L.ipitch, bx, by);
    }
    hipLaunchKernelGGL(k_klt_scharr, gi, dim3(256), 0, s, I + L.ioff, L.w, L.h, L.ipitch, bx, by, D + L.doff, L.dpitch);
  }
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipStreamSynchronize(s));  // the source image may be reused
  k->slot_w[slot] = width, k->slot_h[slot] = height, k->slot_levels[slot] = G.nlevels;
  if (levels_built) *levels_built = G.nlevels;
  return UVO_OK;
}

int uvo_klt_build_pyramid(uvo_klt* k, int slot, const uint8_t* img, int width, int height, ptrdiff_t stride, int* levels_built) {
  if (!k || !img) return fail(UVO_E_BADARG, "null pointer");
  if (slot < 0 || slot >= k->cfg.slots) return fail(UVO_E_BADARG, "slot outside 0..slots-1");
  if (width < 8 || height < 8 || width > k->cfg.max_width || height > k->cfg.max_height || stride < width ||
      (int64_t)width * height > (int64_t)k->cfg.max_width * k->cfg.max_height)
    return fail(UVO_E_BADARG, "image size outside what the handle was sized for");
  UVO_HIP_CHECK(hipSetDevice(k->cfg.device));
  UVO_HIP_CHECK(hipMemcpy2DAsync(k->d_in, width, img, stride, width, (size_t)height, hipMemcpyHostToDevice, k->stream));
  return klt_build_from_device(k, slot, k->d_in, width, height, (int64_t)width, levels_built);
}

// The image is the result of the extractor handle's last uvo_clahe() call, still in HBM: no upload.  The extractor's stream is
// ordered in front of this handle's stream by an event.
extern "C" const uint8_t* uvo_extractor_clahe_internal(uvo_extractor* h, int* width, int* height);
extern "C" hipStream_t uvo_extractor_stream_internal(uvo_extractor* h);
extern "C" int uvo_extractor_device_internal(uvo_extractor* h);
int uvo_klt_build_pyramid_from_extractor(uvo_klt* k, int slot, uvo_extractor* h, int* levels_built) {
  if (!k || !h) return fail(UVO_E_BADARG, "null pointer");
  if (slot < 0 || slot >= k->cfg.slots) return fail(UVO_E_BADARG, "slot outside 0..slots-1");
  if (uvo_extractor_device_internal(h) != k->cfg.device) return fail(UVO_E_BADARG, "handles live on different devices");
  int width = 0, height = 0;
  const uint8_t* d_src = uvo_extractor_clahe_internal(h, &width, &height);
  if (!d_src || width < 8 || height < 8) return fail(UVO_E_BADARG, "the extractor holds no uvo_clahe() result");
  if (width > k->cfg.max_width || height > k->cfg.max_height || (int64_t)width * height > (int64_t)k->cfg.max_width * k->cfg.max_height)
    return fail(UVO_E_BADARG, "image size outside what the handle was sized for");
  UVO_HIP_CHECK(hipSetDevice(k->cfg.device));
  hipEvent_t ev;
  UVO_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e1 = hipEventRecord(ev, uvo_extractor_stream_internal(h));
  hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(k->stream, ev, 0) : e1;
  (void)hipEventDestroy(ev);
  if (e2 != hipSuccess) return fail(UVO_E_HIP, "stream ordering between the extractor and the tracker failed");
  return klt_build_from_device(k, slot, d_src, width, height, (int64_t)width, levels_built);
}

int uvo_klt_read_level(uvo_klt* k, int slot, int level, uint8_t* img, int16_t* deriv, int* width, int* height) {
  if (!k || !width || !height) return fail(UVO_E_BADARG, "null pointer");
  if (slot < 0 || slot >= k->cfg.slots || level < 0 || level >= k->slot_levels[slot]) return fail(UVO_E_BADARG, "no such pyramid level");
  UVO_HIP_CHECK(hipSetDevice(k->cfg.device));
  KltGeom G;
  int64_t ib, db;
  klt_geometry(k->slot_w[slot], k->slot_h[slot], k->cfg.win_width, k->cfg.win_height, k->cfg.max_level, G, &ib, &db);
  const KltLevel& L = G.l[level];
  *width = L.w, *height = L.h;
  const uint8_t* I = k->d_img + (int64_t)slot * k->img_block + L.ioff + (int64_t)G.by * L.ipitch + G.bx;
  const int16_t* D = k->d_der + (int64_t)slot * k->der_block + L.doff + (int64_t)G.by * L.dpitch + 2 * G.bx;
  if (img) UVO_HIP_CHECK(hipMemcpy2D(img, L.w, I, L.ipitch, L.w, L.h, hipMemcpyDeviceToHost));
  if (deriv) UVO_HIP_CHECK(hipMemcpy2D(deriv, (size_t)L.w * 4, D, (size_t)L.dpitch * 2, (size_t)L.w * 4, L.h, hipMemcpyDeviceToHost));
  return UVO_OK;
}

}  // extern "C"

static int check_camera_model(const uvo_camera_model* cam, UndistortCam& C) {
  if (!cam) return fail(UVO_E_BADARG, "null camera model");
  if (!(cam->fx != 0.f) || !(cam->fy != 0.f) || cam->n_dist < 0 || cam->n_dist > 8 || (cam->fisheye && cam->n_dist > 4))
    return fail(UVO_E_BADARG, "bad camera model (focal lengths non-zero, at most 8 distortion coefficients, 4 for the fisheye model)");
  C.fx = (double)cam->fx, C.fy = (double)cam->fy, C.cx = (double)cam->cx, C.cy = (double)cam->cy;  // mK is CV_32F: the values widen
  for (int i = 0; i < 8; ++i) C.k[i] = i < cam->n_dist ? (double)cam->dist[i] : 0.0;
  C.fisheye = cam->fisheye ? 1 : 0;
  return UVO_OK;
}

// the LK step, optionally followed on the device by Tracking::undistort_point of both point sets (one upload, one download)
static int klt_track_impl(uvo_klt* k, int prev_slot, int next_slot, const float* prev_pts, float* next_pts, int n, int max_level, int max_count,
                          double epsilon, double min_eig_threshold, uint8_t* status, float* err, const uvo_camera_model* cam, float* prev_un,
                          float* next_un) {
  if (!k) return fail(UVO_E_BADARG, "null handle");
  if (prev_slot < 0 || prev_slot >= k->cfg.slots || next_slot < 0 || next_slot >= k->cfg.slots || k->slot_levels[prev_slot] == 0 ||
      k->slot_levels[next_slot] == 0)
    return fail(UVO_E_BADARG, "pyramid slot not built");
  if (k->slot_w[prev_slot] != k->slot_w[next_slot] || k->slot_h[prev_slot] != k->slot_h[next_slot])
    return fail(UVO_E_BADARG, "the two pyramids have different sizes");
  if (n < 0 || n > k->cfg.max_points) return fail(UVO_E_BADARG, "point count outside 0..max_points");
  if (n == 0) return UVO_OK;
  if (!prev_pts || !next_pts || !status || !err) return fail(UVO_E_BADARG, "null pointer");
  UndistortCam UC;
  if (cam) {
    if (!prev_un || !next_un) return fail(UVO_E_BADARG, "null pointer");
    const int rc = check_camera_model(cam, UC);
    if (rc) return rc;
  }
  UVO_HIP_CHECK(hipSetDevice(k->cfg.device));
  hipStream_t s = k->stream;
  KltGeom G;
  int64_t ib, db;
  klt_geometry(k->slot_w[prev_slot], k->slot_h[prev_slot], k->cfg.win_width, k->cfg.win_height, k->cfg.max_level, G, &ib, &db);
  max_level = std::min(std::max(max_level, 0), G.nlevels - 1);       // calcOpticalFlowPyrLK: maxLevel = min(levels of both pyramids)
  max_count = std::min(std::max(max_count, 0), 100);                   // criteria.maxCount clamp
  epsilon = std::min(std::max(epsilon, 0.), 10.);
  epsilon *= epsilon;
  // layout of the block for this call: prev [n][2] f32 | next [n][2] f32 | err [n] f32 | status [n] u8
  const size_t N = (size_t)n, o_next = N * 8, o_err = N * 16, o_status = N * 20;
  k->d_prev = reinterpret_cast<float*>(k->d_pts), k->d_next = reinterpret_cast<float*>(k->d_pts + o_next);
  k->d_err = reinterpret_cast<float*>(k->d_pts + o_err), k->d_status = k->d_pts + o_status;
  std::memcpy(k->h_pts, prev_pts, N * 8);
  std::memcpy(k->h_pts + o_next, next_pts, N * 8);
  UVO_HIP_CHECK(hipMemcpyAsync(k->d_pts, k->h_pts, N * 16, hipMemcpyHostToDevice, s));
  const uint8_t* I0 = k->d_img + (int64_t)prev_slot * k->img_block;
  const int16_t* D0 = k->d_der + (int64_t)prev_slot * k->der_block;
  const uint8_t* I1 = k->d_img + (int64_t)next_slot * k->img_block;
  const int npx = k->cfg.win_width * k->cfg.win_height;
  const dim3 grid((n + 3) / 4);
  if (npx <= 448)
    hipLaunchKernelGGL(k_klt_track<7>, grid, dim3(256), 0, s, G, I0, D0, I1, k->d_prev, k->d_next, n, k->cfg.win_width, k->cfg.win_height, max_level,
                       max_count, (float)epsilon, (float)min_eig_threshold, k->d_status, k->d_err);
  else
    hipLaunchKernelGGL(k_klt_track<16>, grid, dim3(256), 0, s, G, I0, D0, I1, k->d_prev, k->d_next, n, k->cfg.win_width, k->cfg.win_height, max_level,
                       max_count, (float)epsilon, (float)min_eig_threshold, k->d_status, k->d_err);
  // undistorted copies behind the status bytes (8-byte aligned): prev_un [n][2] | next_un [n][2]
  const size_t o_un = (N * 21 + 7) & ~(size_t)7;
  if (cam) {
    float* d_un = reinterpret_cast<float*>(k->d_pts + o_un);
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, s, UC, k->d_prev, n, d_un);
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, s, UC, k->d_next, n, d_un + 2 * N);
  }
  UVO_HIP_CHECK(hipGetLastError());
  const size_t down = (cam ? o_un + N * 16 : N * 21) - o_next;
  UVO_HIP_CHECK(hipMemcpyAsync(k->h_pts + o_next, k->d_pts + o_next, down, hipMemcpyDeviceToHost, s));  // next, err, status (, prev_un, next_un)
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  std::memcpy(next_pts, k->h_pts + o_next, N * 8);
  std::memcpy(err, k->h_pts + o_err, N * 4);
  std::memcpy(status, k->h_pts + o_status, N);
  if (cam) {
    std::memcpy(prev_un, k->h_pts + o_un, N * 8);
    std::memcpy(next_un, k->h_pts + o_un + N * 8, N * 8);
  }
  return UVO_OK;
}

extern "C" {

int uvo_klt_track(uvo_klt* k, int prev_slot, int next_slot, const float* prev_pts, float* next_pts, int n, int max_level, int max_count,
                  double epsilon, double min_eig_threshold, uint8_t* status, float* err) {
  return klt_track_impl(k, prev_slot, next_slot, prev_pts, next_pts, n, max_level, max_count, epsilon, min_eig_threshold, status, err, nullptr, nullptr,
                        nullptr);
}

int uvo_klt_track_undistorted(uvo_klt* k, int prev_slot, int next_slot, const float* prev_pts, float* next_pts, int n, int max_level, int max_count,
                              double epsilon, double min_eig_threshold, const uvo_camera_model* cam, uint8_t* status, float* err, float* prev_un,
                              float* next_un) {
  if (!cam) return fail(UVO_E_BADARG, "null camera model");
  return klt_track_impl(k, prev_slot, next_slot, prev_pts, next_pts, n, max_level, max_count, epsilon, min_eig_threshold, status, err, cam, prev_un,
                        next_un);
}

int uvo_undistort_points(uvo_klt* k, const uvo_camera_model* cam, const float* pts, int n, float* out) {
  if (!k) return fail(UVO_E_BADARG, "null handle");
  if (n < 0 || n > 2 * k->cfg.max_points) return fail(UVO_E_BADARG, "point count outside 0..2*max_points");
  if (n == 0) return UVO_OK;
  if (!pts || !out) return fail(UVO_E_BADARG, "null pointer");
  UndistortCam UC;
  const int rc = check_camera_model(cam, UC);
  if (rc) return rc;
  UVO_HIP_CHECK(hipSetDevice(k->cfg.device));
  hipStream_t s = k->stream;
  const size_t N = (size_t)n;
  std::memcpy(k->h_pts, pts, N * 8);
  UVO_HIP_CHECK(hipMemcpyAsync(k->d_pts, k->h_pts, N * 8, hipMemcpyHostToDevice, s));
  float* d_out = reinterpret_cast<float*>(k->d_pts + N * 8);
  hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, s, UC, reinterpret_cast<const float*>(k->d_pts), n, d_out);
  UVO_HIP_CHECK(hipGetLastError());
  UVO_HIP_CHECK(hipMemcpyAsync(k->h_pts + N * 8, d_out, N * 8, hipMemcpyDeviceToHost, s));
  UVO_HIP_CHECK(hipStreamSynchronize(s));
  std::memcpy(out, k->h_pts + N * 8, N * 8);
  return UVO_OK;
}


}  // extern "C"
