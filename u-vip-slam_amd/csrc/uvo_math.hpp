// Scalar numerics shared by the HIP kernels and the host-side table builders.
//
// Everything the reference delegates to libm / OpenCV scalar helpers on the descriptor and orientation
// path is written out here once, in IEEE operations that give the same bits on the host and on gfx950
// (the library is built with -ffp-contract=off, and HIP's fp32 divide is correctly rounded):
//   * uvo_sincosf  : glibc >= 2.28 sinf/cosf (sysdeps/ieee754/flt-32/s_sincosf.h) -- what
//                    `cos(angle)` / `sin(angle)` at src/ORBextractor.cc:160-161 resolve to (float overloads).
//                    tests/test_math_host.py checks it bit-for-bit against the platform libm over every
//                    float in [0, 2*pi].
//   * uvo_fast_atan2: OpenCV 3.4 cv::fastAtan2 (atan_f32), call site src/ORBextractor.cc:151.
//   * uvo_logf     : glibc >= 2.27 logf (sysdeps/ieee754/flt-32/e_logf.c, the ARM optimized-routines algorithm:
//                    16-entry table, degree-3 polynomial in double) -- what `log(ratio)` in MapPoint::PredictScale
//                    (src/MapPoint.cc:381) resolves to (float overload through the global using-directive of
//                    include/cluster.h:17).  Checked bit-for-bit against the platform libm in tests/test_math_host.py.
//   * uvo_cv_round  : cvRound = round-half-to-even (cvtss2si), src/ORBextractor.cc:129,163,167-168.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UVO_HD __host__ __device__ inline
#else
#define UVO_HD inline
#endif

namespace uvo {

UVO_HD int cv_round(float v) { return (int)__builtin_rintf(v); }

UVO_HD uint32_t f32_abstop12(float f) {
  union {
    float f;
    uint32_t u;
  } c;
  c.f = f;
  return (c.u >> 20) & 0x7ff;
}

// polynomial selected by quadrant parity; `neg` selects the second coefficient set (cosine terms negated)
UVO_HD float sincosf_poly(double x, double x2, bool neg, int n) {
  const double c0 = neg ? -0x1p0 : 0x1p0;
  const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
  const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
  const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
  const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double t1 = s2 + x2 * s3;
    double x7 = x3 * x2;
    double s = x + x3 * s1;
    return (float)(s + x7 * t1);
  } else {
    double x4 = x2 * x2;
    double t2 = c3 + x2 * c4;
    double t1 = c0 + x2 * c1;
    double x6 = x4 * x2;
    double c = t1 + x4 * c2;
    return (float)(c + x6 * t2);
  }
}

// valid for |y| < 120 (the extractor only ever passes [0, 2*pi])
UVO_HD void uvo_sincosf(float y, float* sinp, float* cosp) {
  double x = y;
  if (f32_abstop12(y) < f32_abstop12(0x1.921FB6p-1f)) {  // |y| < pi/4
    double x2 = x * x;
    if (f32_abstop12(y) < f32_abstop12(0x1p-12f)) {
      *sinp = y;
      *cosp = 1.0f;
      return;
    }
    *sinp = sincosf_poly(x, x2, false, 0);
    *cosp = sincosf_poly(x, x2, false, 1);
    return;
  }
  const double hpi_inv = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  const double hpi = 0x1.921FB54442D18p0;
  double r = x * hpi_inv;
  int n = ((int32_t)r + 0x800000) >> 24;
  x = x - n * hpi;
  const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  const bool neg = (n & 2) != 0;
  *sinp = sincosf_poly(x * sgn, x * x, neg, n);
  *cosp = sincosf_poly(x * sgn, x * x, neg, n ^ 1);
}

UVO_HD float uvo_fast_atan2(float y, float x) {
  const float k = 57.295779513082320877f;  // (float)(180/CV_PI)
  const float p1 = 0.9997878412794807f * k;
  const float p3 = -0.3258083974640975f * k;
  const float p5 = 0.1555786518463281f * k;
  const float p7 = -0.04432655554792128f * k;
  const float eps = 2.2204460492503131e-16f;  // (float)DBL_EPSILON
  float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// glibc logf for finite positive normal x (the only inputs PredictScale produces: a ratio of two positive distances);
// zero, negative, subnormal, inf and nan inputs take glibc's special-case branch, reproduced for completeness.
UVO_HD float uvo_logf(float x) {
  const double T_invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                             0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                             0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
  const double T_logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                             -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                             -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  union {
    float f;
    uint32_t u;
  } c;
  c.f = x;
  uint32_t ix = c.u;
  if (ix == 0x3f800000u) return 0.f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return -__builtin_inff();         // log(+-0) = -inf
    if (ix == 0x7f800000u) return x;                    // log(inf) = inf
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");  // negative or nan
    c.f = x * 0x1p23f;                                  // subnormal: normalise
    ix = c.u - (23u << 23);
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) % 16u);
  const int k = (int32_t)tmp >> 23;
  c.u = ix - (tmp & (0x1ffu << 23));
  const double z = (double)c.f;
  const double r = z * T_invc[i] - 1;
  const double y0 = T_logc[i] + (double)k * Ln2;
  const double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

}  // namespace uvo
