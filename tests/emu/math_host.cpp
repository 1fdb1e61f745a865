// Host build of the product's scalar numerics header (u-vip-slam_amd/csrc/uvo_math.hpp) so that the exact code the
// HIP kernels run can be compared with the platform libm / the oracle on a machine without a GPU.  Test scaffolding.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../u-vip-slam_amd/csrc/uvo_math.hpp"

extern "C" {
void mh_sincosf(float a, float* s, float* c) { uvo::uvo_sincosf(a, s, c); }
float mh_fast_atan2(float y, float x) { return uvo::uvo_fast_atan2(y, x); }
int mh_cv_round(float v) { return uvo::cv_round(v); }
float mh_logf(float v) { return uvo::uvo_logf(v); }
// number of floats with bit patterns in [lo_bits, hi_bits] where uvo_logf differs from libm logf (nan == nan)
long mh_logf_mismatches(uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint32_t* first_bad) {
  std::vector<long> bad(nthreads, 0);
  std::vector<uint32_t> first(nthreads, 0xffffffffu);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      for (uint64_t u = (uint64_t)lo_bits + t; u <= hi_bits; u += nthreads) {
        uint32_t uu = (uint32_t)u;
        float y;
        memcpy(&y, &uu, 4);
        const float a = uvo::uvo_logf(y), b = logf(y);
        if (memcmp(&a, &b, 4) && !(a != a && b != b)) {
          if (!bad[t]) first[t] = uu;
          ++bad[t];
        }
      }
    });
  for (auto& x : th) x.join();
  long n = 0;
  uint32_t fb = 0xffffffffu;
  for (int t = 0; t < nthreads; ++t) n += bad[t], fb = first[t] < fb ? first[t] : fb;
  if (first_bad) *first_bad = fb;
  return n;
}
// number of floats in [lo_bits, hi_bits] (as IEEE bit patterns of non-negative floats) where uvo_sincosf differs from libm sinf/cosf
long mh_sincos_mismatches(uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint32_t* first_bad) {
  std::vector<long> bad(nthreads, 0);
  std::vector<uint32_t> first(nthreads, 0xffffffffu);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      for (uint64_t u = (uint64_t)lo_bits + t; u <= hi_bits; u += nthreads) {
        uint32_t uu = (uint32_t)u;
        float y, s, c;
        memcpy(&y, &uu, 4);
        uvo::uvo_sincosf(y, &s, &c);
        float rs = sinf(y), rc = cosf(y);
        if (memcmp(&s, &rs, 4) || memcmp(&c, &rc, 4)) {
          if (!bad[t]) first[t] = uu;
          ++bad[t];
        }
      }
    });
  for (auto& x : th) x.join();
  long n = 0;
  uint32_t fb = 0xffffffffu;
  for (int t = 0; t < nthreads; ++t) n += bad[t], fb = first[t] < fb ? first[t] : fb;
  if (first_bad) *first_bad = fb;
  return n;
}
}
