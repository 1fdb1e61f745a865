// Micro-benchmark: issue rate of the packed fp32 instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two fp32 results per lane) against
// v_fma_f32 / v_add_f32, and of v_cvt_f32_u32 / v_cvt_pk_u8_f32 / v_fract_f32 -- the instructions of the blur's column pass (gauss_body.hpp).
//   hipcc --offload-arch=gfx950 -O2 -o pk_f32_rate pk_f32_rate.hip && ./pk_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
  f2 a[8];
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{a0 + threadIdx.x * (i + 1), a0 * 0.5f + i}, s[i] = a0 + i + threadIdx.x;
  const f2 b{1.0001f, 0.9999f}, c{0.25f, 0.125f};
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[r % 8]) : "v"(b.x), "v"(c.x));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[r % 8]) : "v"(b), "v"(c));
      if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[r % 8]) : "v"(c.x));
      if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[r % 8]) : "v"(c));
      if (OP == 4) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(s[r % 8]));
      if (OP == 5) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(acc) : "v"(s[r % 8]));
      if (OP == 6) asm volatile("v_fract_f32 %0, %0" : "+v"(s[r % 8]));
      if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[r % 8]) : "v"(b));
      if (OP == 8) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(acc) : "v"(__builtin_bit_cast(uint32_t, s[r % 8])), "v"(0x01020304u));
      if (OP == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[r % 8]) : "v"(b.x), "v"(c.x));            // VOP2: D += S0 * S1, two vector sources
      if (OP == 10) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[r % 8]) : "s"(a0), "v"(c.x));             // ... one of them scalar (the blur's tap)
      if (OP == 11) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[r % 8]) : "s"(a0), "v"(c.x));          // VOP3 with the same operands
      if (OP == 12) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(s[r % 8]) : "v"(b.x));
      if (OP == 13) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc) : "v"(__builtin_bit_cast(uint32_t, s[r % 8])), "v"(0x37u));
      if (OP == 14) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(acc) : "v"(__builtin_bit_cast(uint32_t, s[r % 8])), "v"(0x00370031u));
    }
  }
  float t = (float)acc;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += a[i].x + a[i].y + s[i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int OP>
void run(const char* name, int waves_per_simd) {
  float* d;
  const int blocks = 256 * waves_per_simd, iters = 4000;
  (void)hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(d, 100, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  k<OP><<<blocks, 256>>>(d, iters, 1.f);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("%-16s waves/SIMD %d : %6.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 32));
  (void)hipFree(d);
}
int main() {
  for (int w : {1, 4, 8}) {
    run<0>("v_fma_f32", w), run<1>("v_pk_fma_f32", w), run<2>("v_add_f32", w), run<3>("v_pk_add_f32", w), run<7>("v_pk_mul_f32", w);
    run<4>("v_cvt_f32_u32", w), run<5>("v_cvt_pk_u8_f32", w), run<6>("v_fract_f32", w), run<8>("v_dot4_u32_u8", w);
    run<9>("v_fmac_f32 vv", w), run<10>("v_fmac_f32 sv", w), run<11>("v_fma_f32 svv", w), run<12>("v_mul_f32", w), run<13>("v_mad_u32_u24", w), run<14>("v_dot2_u32_u16", w);
  }
  return 0;
}
