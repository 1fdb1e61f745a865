// Micro-benchmark 2: issue cost of the byte/packed instructions considered for the stencil kernels (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short2_t __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t a0) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x * (i + 1) * 2654435761u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t b = a[(i + 1) & 7], c = a[(i + 3) & 7];
      if (OP == 0) a[i] = __builtin_amdgcn_udot4(a[i], b, c, false);
      if (OP == 1) a[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a[i]), __builtin_bit_cast(short2_t, b), (int)c, false);
      if (OP == 2) a[i] = __builtin_amdgcn_perm(a[i], b, 0x0c010c00u) + c;
      if (OP == 3) a[i] = __builtin_amdgcn_alignbyte(a[i], b, c & 3);
      if (OP == 4) { typedef unsigned short us2 __attribute__((ext_vector_type(2))); us2 x = __builtin_bit_cast(us2, a[i]), y = __builtin_bit_cast(us2, b); us2 z = __builtin_elementwise_min(x, y); a[i] = __builtin_bit_cast(uint32_t, z) ^ c; }
      if (OP == 5) a[i] = __builtin_amdgcn_ubfe(a[i], 8, 8) + b;
      if (OP == 6) a[i] = (a[i] & 0xff) * (b & 0xfff) + c;   // mad_u32_u24 candidates
      if (OP == 7) a[i] = __builtin_amdgcn_sad_u8(a[i], b, c);
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int ops) {
  uint32_t* d;
  const int blocks = 256 * 8, iters = 20000;
  (void)hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(d, 100, 1);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  k<OP><<<blocks, 256>>>(d, iters, 1);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double wi = (double)blocks * 4 * iters * 8 * ops;
  printf("%-34s %8.3f ms  %7.1f wave-instr/ns = %5.2f cycles/instr/SIMD @2.4GHz (assuming %d instr per op)\n", name, ms, wi / (ms * 1e6),
         1024.0 * 2.4 / (wi / (ms * 1e6)), ops);
  (void)hipFree(d);
}
int main() {
  run<0>("v_dot4_u32_u8", 1);
  run<1>("v_dot2_i32_i16", 1);
  run<2>("v_perm_b32 + add", 2);
  run<3>("v_alignbyte_b32 (+and)", 2);
  run<4>("v_pk_min_u16 + xor", 2);
  run<5>("v_bfe_u32 + add", 2);
  run<6>("and,and,mad24", 3);
  run<7>("v_sad_u8", 1);
  return 0;
}
