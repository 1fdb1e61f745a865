// ORACLE -- TEST INFRASTRUCTURE ONLY (see orb_oracle.hpp header).  PARITY UNPINNED.
//
// The KLT step that runs on every frame in front of the extractor:
//   cv::buildOpticalFlowPyramid(im, imgpyr, mWin_Size, mPyr_Levels)                              src/FrameKTL.cc:76
//   cv::calcOpticalFlowPyrLK(img0pyr, img1pyr, pts0, pts1, mask_klt, error, win_size, pyr_levels,
//        TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW + OPTFLOW_LK_GET_MIN_EIGENVALS)   src/Tracking.cc:1046-1047
// restated from OpenCV 3.4 video/src/lkpyramid.cpp (calcSharrDeriv, LKTrackerInvoker, scalar C++ path) and
// imgproc/src/pyramids.cpp (pyrDown, 8-bit: 1-4-6-4-1 taps, (sum + 128) >> 8) [OCV-RECALL].  The float accumulators of the
// tracker are summed in raster order over the window here (the scalar path); a SIMD OpenCV build sums four partial lanes, so
// positions agree with any real build only to float rounding -- the GPU parity tests use a tolerance for the same reason.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "orb_oracle.hpp"

namespace orc {

static inline int refl101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// pyrDown, CV_8UC1, BORDER_REFLECT_101: dst size ((w+1)/2, (h+1)/2)
static void pyr_down(const uint8_t* src, int sw, int sh, ptrdiff_t sstep, uint8_t* dst, int dw, int dh, ptrdiff_t dstep) {
  std::vector<int> row(dw);
  std::vector<std::vector<int>> rows(5, std::vector<int>(dw));
  for (int y = 0; y < dh; ++y) {
    for (int k = 0; k < 5; ++k) {
      const uint8_t* s = src + (ptrdiff_t)refl101(2 * y + k - 2, sh) * sstep;
      for (int x = 0; x < dw; ++x) {
        const int x0 = refl101(2 * x - 2, sw), x1 = refl101(2 * x - 1, sw), x2 = refl101(2 * x, sw), x3 = refl101(2 * x + 1, sw),
                  x4 = refl101(2 * x + 2, sw);
        rows[k][x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
      }
    }
    for (int x = 0; x < dw; ++x)
      dst[(ptrdiff_t)y * dstep + x] = (uint8_t)((rows[2][x] * 6 + (rows[1][x] + rows[3][x]) * 4 + rows[0][x] + rows[4][x] + 128) >> 8);
  }
}

// calcSharrDeriv: dst = interleaved (dI/dx, dI/dy) int16, 3-10-3 Scharr, REFLECT_101 inside the image itself
static void scharr_deriv(const uint8_t* src, int w, int h, ptrdiff_t sstep, int16_t* dst, ptrdiff_t dstep /* in shorts */) {
  std::vector<int> trow0(w + 2), trow1(w + 2);
  for (int y = 0; y < h; ++y) {
    const uint8_t* srow0 = src + (ptrdiff_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * sstep;
    const uint8_t* srow1 = src + (ptrdiff_t)y * sstep;
    const uint8_t* srow2 = src + (ptrdiff_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * sstep;
    int* t0 = trow0.data() + 1;
    int* t1 = trow1.data() + 1;
    for (int x = 0; x < w; ++x) {
      t0[x] = (srow0[x] + srow2[x]) * 3 + srow1[x] * 10;
      t1[x] = srow2[x] - srow0[x];
    }
    const int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
    t0[-1] = t0[x0], t0[w] = t0[x1];
    t1[-1] = t1[x0], t1[w] = t1[x1];
    int16_t* drow = dst + (ptrdiff_t)y * dstep;
    for (int x = 0; x < w; ++x) {
      drow[2 * x] = (int16_t)(t0[x + 1] - t0[x - 1]);
      drow[2 * x + 1] = (int16_t)((t1[x + 1] + t1[x - 1]) * 3 + t1[x] * 10);
    }
  }
}

void KltPyramid::build(const uint8_t* img, int w, int h, ptrdiff_t stride, int win_w, int win_h, int maxLevel) {
  bx = win_w, by = win_h;
  levels.clear();
  int lw = w, lh = h;
  std::vector<uint8_t> prev;
  for (int l = 0; l <= maxLevel; ++l) {
    Level L;
    L.w = lw, L.h = lh;
    L.istep = lw + 2 * bx;
    L.img.assign((size_t)L.istep * (lh + 2 * by), 0);
    std::vector<uint8_t> cur((size_t)lw * lh);
    if (l == 0) {
      for (int y = 0; y < lh; ++y) memcpy(&cur[(size_t)y * lw], img + (ptrdiff_t)y * stride, lw);
    } else {
      pyr_down(prev.data(), levels[l - 1].w, levels[l - 1].h, levels[l - 1].w, cur.data(), lw, lh, lw);
    }
    // image with a winSize border, BORDER_REFLECT_101 (pyrBorder default)
    copy_make_border_reflect101(View{cur.data(), lw, lh, lw}, L.img.data(), L.istep, by, by, bx, bx);
    // derivatives with a zero border (derivBorder default BORDER_CONSTANT)
    L.dstep = 2 * (lw + 2 * bx);
    L.deriv.assign((size_t)L.dstep * (lh + 2 * by), 0);
    scharr_deriv(cur.data(), lw, lh, lw, L.deriv.data() + (size_t)by * L.dstep + 2 * bx, L.dstep);
    levels.push_back(std::move(L));
    prev.swap(cur);
    lw = (lw + 1) / 2, lh = (lh + 1) / 2;
    if (lw <= win_w || lh <= win_h) break;  // buildOpticalFlowPyramid stops when the next level would not exceed the window
  }
}

static inline int cv_floor(float v) { return (int)floorf(v); }
static inline int cv_round(float v) { return (int)lrintf(v); }
#define KLT_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// The window sums (iA11.., ib1, ib2) are float accumulations, so their value depends on the order of the additions.  sum_mode 0 is the
// raster order of OpenCV's generic C++ loop (what this oracle stands for); sum_mode 1 is the order the HIP kernel adds in (window pixel
// idx goes to lane idx % 64, every lane adds its pixels in increasing idx, then a 64-lane xor butterfly: off = 32, 16, .., 1) -- a
// diagnostic mode: the kernel must equal it bit for bit, and the difference between the two modes is a property of the algorithm that
// the CPU suite quantifies (tests/test_oracle_kat.py).  OpenCV's own SIMD builds add in yet another order.
namespace {
struct WinSum {
  int mode, idx = 0;
  float seq = 0.f, lane[64];
  explicit WinSum(int m) : mode(m) {
    for (float& v : lane) v = 0.f;
  }
  void add(float v) {
    if (mode == 0)
      seq += v;
    else
      lane[idx & 63] += v;
    ++idx;
  }
  float total() {
    if (mode == 0) return seq;
    for (int off = 32; off > 0; off >>= 1) {
      float t[64];
      for (int l = 0; l < 64; ++l) t[l] = lane[l] + lane[l ^ off];
      for (int l = 0; l < 64; ++l) lane[l] = t[l];
    }
    return lane[0];
  }
};
inline void note(float* margin, int i, double m) {
  if (margin && m < margin[i]) margin[i] = (float)m;
}
}  // namespace

// margin (optional, per point): the smallest relative distance of any yes/no decision taken for the point to its threshold -- the
// minimum-eigenvalue and determinant tests, the image-bounds tests (in pixels), the two termination tests.  A run whose margin is
// large cannot change its decisions under a perturbation of the sums' last bits.
void klt_track(const KltPyramid& P0, const KltPyramid& P1, const float* prevPts, float* nextPts, int npts, int win_w, int win_h, int maxLevel,
               int maxCount, double epsilon, double minEigThreshold, uint8_t* status, float* err, int sum_mode, float* margin) {
  if (margin)
    for (int i = 0; i < npts; ++i) margin[i] = 1e30f;
  maxLevel = std::min(maxLevel, (int)std::min(P0.levels.size(), P1.levels.size()) - 1);
  maxCount = std::min(std::max(maxCount, 0), 100);
  epsilon = std::min(std::max(epsilon, 0.), 10.);
  epsilon *= epsilon;
  for (int i = 0; i < npts; ++i) status[i] = 1, err[i] = 0;
  const float halfx = (win_w - 1) * 0.5f, halfy = (win_h - 1) * 0.5f;
  std::vector<int16_t> IWin((size_t)win_w * win_h), dIWin((size_t)win_w * win_h * 2);
  for (int level = maxLevel; level >= 0; --level) {
    const KltPyramid::Level& I = P0.levels[level];
    const KltPyramid::Level& J = P1.levels[level];
    const uint8_t* Ibase = I.img.data() + (size_t)P0.by * I.istep + P0.bx;
    const int16_t* Dbase = I.deriv.data() + (size_t)P0.by * I.dstep + 2 * P0.bx;
    const uint8_t* Jbase = J.img.data() + (size_t)P1.by * J.istep + P1.bx;
    for (int ptidx = 0; ptidx < npts; ++ptidx) {
      float prevx = prevPts[2 * ptidx] * (float)(1. / (1 << level)), prevy = prevPts[2 * ptidx + 1] * (float)(1. / (1 << level));
      float nextx, nexty;
      if (level == maxLevel) {
        nextx = nextPts[2 * ptidx] * (float)(1. / (1 << level)), nexty = nextPts[2 * ptidx + 1] * (float)(1. / (1 << level));  // USE_INITIAL_FLOW
      } else {
        nextx = nextPts[2 * ptidx] * 2.f, nexty = nextPts[2 * ptidx + 1] * 2.f;
      }
      nextPts[2 * ptidx] = nextx, nextPts[2 * ptidx + 1] = nexty;
      prevx -= halfx, prevy -= halfy;
      int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
      note(margin, ptidx, std::min(std::min(std::fabs(prevx + win_w), std::fabs(prevx - I.w)), std::min(std::fabs(prevy + win_h), std::fabs(prevy - I.h))));
      if (ipx < -win_w || ipx >= I.w || ipy < -win_h || ipy >= I.h) {
        if (level == 0) status[ptidx] = 0, err[ptidx] = 0;
        continue;
      }
      float a = prevx - ipx, b = prevy - ipy;
      const int W_BITS = 14, W_BITS1 = 14;
      const float FLT_SCALE = 1.f / (1 << 20);
      int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
      int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
      int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      const ptrdiff_t stepI = I.istep, dstep = I.dstep, stepJ = J.istep;
      WinSum sA11(sum_mode), sA12(sum_mode), sA22(sum_mode);
      for (int y = 0; y < win_h; y++) {
        const uint8_t* src = Ibase + (ptrdiff_t)(y + ipy) * stepI + ipx;
        const int16_t* dsrc = Dbase + (ptrdiff_t)(y + ipy) * dstep + ipx * 2;
        int16_t* Iptr = &IWin[(size_t)y * win_w];
        int16_t* dIptr = &dIWin[(size_t)y * win_w * 2];
        for (int x = 0; x < win_w; x++, dsrc += 2, dIptr += 2) {
          int ival = KLT_DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11, W_BITS1 - 5);
          int ixval = KLT_DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11, W_BITS1);
          int iyval = KLT_DESCALE(dsrc[1] * iw00 + dsrc[2 + 1] * iw01 + dsrc[dstep + 1] * iw10 + dsrc[dstep + 2 + 1] * iw11, W_BITS1);
          Iptr[x] = (int16_t)ival;
          dIptr[0] = (int16_t)ixval;
          dIptr[1] = (int16_t)iyval;
          sA11.add((float)(ixval * ixval));
          sA12.add((float)(ixval * iyval));
          sA22.add((float)(iyval * iyval));
        }
      }
      float A11 = sA11.total() * FLT_SCALE, A12 = sA12.total() * FLT_SCALE, A22 = sA22.total() * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win_w * win_h);
      err[ptidx] = (float)minEig;  // OPTFLOW_LK_GET_MIN_EIGENVALS
      note(margin, ptidx, std::fabs((double)minEig - minEigThreshold) / minEigThreshold);
      note(margin, ptidx, std::fabs((double)D - 1.1920929e-07) / 1.1920929e-07);
      if (minEig < minEigThreshold || D < 1.1920929e-07f /* FLT_EPSILON */) {
        if (level == 0) status[ptidx] = 0;
        continue;
      }
      D = 1.f / D;
      nextx -= halfx, nexty -= halfy;
      float pdx = 0, pdy = 0;
      for (int j = 0; j < maxCount; j++) {
        int inx = cv_floor(nextx), iny = cv_floor(nexty);
        note(margin, ptidx, std::min(std::min(std::fabs(nextx + win_w), std::fabs(nextx - J.w)), std::min(std::fabs(nexty + win_h), std::fabs(nexty - J.h))));
        if (inx < -win_w || inx >= J.w || iny < -win_h || iny >= J.h) {
          if (level == 0) status[ptidx] = 0;
          break;
        }
        a = nextx - inx, b = nexty - iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        WinSum sb1(sum_mode), sb2(sum_mode);
        for (int y = 0; y < win_h; y++) {
          const uint8_t* Jptr = Jbase + (ptrdiff_t)(y + iny) * stepJ + inx;
          const int16_t* Iptr = &IWin[(size_t)y * win_w];
          const int16_t* dIptr = &dIWin[(size_t)y * win_w * 2];
          for (int x = 0; x < win_w; x++, dIptr += 2) {
            int diff = KLT_DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 + Jptr[x + stepJ + 1] * iw11, W_BITS1 - 5) - Iptr[x];
            sb1.add((float)(diff * dIptr[0]));
            sb2.add((float)(diff * dIptr[1]));
          }
        }
        float b1 = sb1.total() * FLT_SCALE, b2 = sb2.total() * FLT_SCALE;
        float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
        nextx += dx, nexty += dy;
        nextPts[2 * ptidx] = nextx + halfx, nextPts[2 * ptidx + 1] = nexty + halfy;
        if (epsilon > 0) note(margin, ptidx, std::fabs((double)dx * dx + (double)dy * dy - epsilon) / epsilon);
        if (j > 0) note(margin, ptidx, std::min(std::fabs(std::fabs((double)dx + pdx) - 0.01), std::fabs(std::fabs((double)dy + pdy) - 0.01)) / 0.01);
        if ((double)dx * dx + (double)dy * dy <= epsilon) break;  // delta.ddot(delta)
        if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
          nextPts[2 * ptidx] -= dx * 0.5f, nextPts[2 * ptidx + 1] -= dy * 0.5f;
          break;
        }
        pdx = dx, pdy = dy;
      }
    }
  }
}

// Tracking::undistort_point (src/Tracking.cc:1265-1283): cv::undistortPoints(src, dst, K, D, noArray(), K) for the pin-hole model,
// cv::fisheye::undistortPoints(src, dst, K, D, Mat(), K) when Fisheye_Cam.  [OCV-RECALL, OpenCV 3.4.6, unpinned]
//   imgproc/undistort.cpp cvUndistortPointsInternal: everything in double; x = (u - cx) * (1 / fx); five iterations (criteria =
//   (ITER, 5, 0.01): no epsilon test) of x = (x0 - deltaX) * icdist with the Brown model's radial quotient and tangential terms; then
//   the re-projection with P * R = K as a full 3x3 product (the zero entries stay in the sums) and the division by w.
//   calib3d/fisheye.cpp undistortPoints: pw = (p - c) / f; theta_d = |pw| clamped to [-pi/2, pi/2]; above 1e-8 at most ten Newton steps
//   on theta * (1 + k1 theta^2 + .. + k4 theta^8) = theta_d, ended by |step| < 1e-8; scale = tan(theta) / theta_d; re-projection
//   with K, division by the third component.
void undistort_points(const float* pts, int n, float fxf, float fyf, float cxf, float cyf, const float* dist, int n_dist, bool fisheye, float* out) {
  const double fx = fxf, fy = fyf, cx = cxf, cy = cyf;
  double k[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n_dist && i < 8; ++i) k[i] = dist[i];
  for (int i = 0; i < n; ++i) {
    if (!fisheye) {
      const double ifx = 1. / fx, ify = 1. / fy;
      double x = pts[2 * i], y = pts[2 * i + 1];
      x = (x - cx) * ifx;
      y = (y - cy) * ify;
      const double x0 = x, y0 = y;
      for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
      }
      const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
      double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
      double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
      double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
      out[2 * i] = (float)(xx * ww), out[2 * i + 1] = (float)(yy * ww);
    } else {
      const double pi0 = pts[2 * i], pi1 = pts[2 * i + 1];
      const double pw0 = (pi0 - cx) / fx, pw1 = (pi1 - cy) / fy;
      double scale = 1.0;
      double theta_d = std::sqrt(pw0 * pw0 + pw1 * pw1);
      const double CV_PI_ = 3.1415926535897932384626433832795;
      theta_d = std::min(std::max(-CV_PI_ / 2., theta_d), CV_PI_ / 2.);
      if (theta_d > 1e-8) {
        double theta = theta_d;
        const double EPS = 1e-8;
        for (int j = 0; j < 10; j++) {
          double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
          double k0_theta2 = k[0] * theta2, k1_theta4 = k[1] * theta4, k2_theta6 = k[2] * theta6, k3_theta8 = k[3] * theta8;
          double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                             (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
          theta = theta - theta_fix;
          if (std::fabs(theta_fix) < EPS) break;
        }
        scale = std::tan(theta) / theta_d;
      }
      const double pu0 = pw0 * scale, pu1 = pw1 * scale;
      const double pr0 = fx * pu0 + 0. * pu1 + cx * 1.0, pr1 = 0. * pu0 + fy * pu1 + cy * 1.0, pr2 = 0. * pu0 + 0. * pu1 + 1. * 1.0;
      out[2 * i] = (float)(pr0 / pr2), out[2 * i + 1] = (float)(pr1 / pr2);
    }
  }
}

}  // namespace orc
