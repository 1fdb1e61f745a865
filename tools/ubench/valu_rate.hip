// Micro-benchmark: sustained VALU issue rate of the integer ops the FAST / blur kernels are made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters, int a0) {
  int a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x * (i + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = a[i] + a[(i + 1) & 7];                                   // v_add_u32
      if (OP == 1) a[i] = min(a[i], min(a[(i + 1) & 7], a[(i + 3) & 7] ^ it));      // v_min3_i32 (+xor)
      if (OP == 2) a[i] = __mul24(a[i], 3) + a[(i + 1) & 7];                        // v_mad_i32_i24
      if (OP == 3) a[i] = a[i] * a[(i + 1) & 7];                                    // v_mul_lo_u32
      if (OP == 4) a[i] = ((a[i] >> 8) & 0xff) + a[(i + 1) & 7];                    // v_bfe + add (or sdwa)
      if (OP == 5) a[i] = (a[i] > a[(i + 1) & 7]) ? a[(i + 2) & 7] : a[i] + 1;      // v_cmp + v_cndmask
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int ops_per_iter_per_lane) {
  int* d;
  const int blocks = 256 * 8, iters = 20000;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(d, 100, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<blocks, 256>>>(d, iters, 1);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double waveinstr = (double)blocks * 4 * iters * 8 * ops_per_iter_per_lane;
  printf("%-28s %8.3f ms  %7.1f wave-instr/ns chip  = %5.2f cycles per wave-instr per SIMD @2.4GHz\n", name, ms, waveinstr / (ms * 1e6),
         1024.0 * 2.4 / (waveinstr / (ms * 1e6)));
  hipFree(d);
}

int main() {
  run<0>("v_add_u32", 1);
  run<1>("v_min3_i32 + v_xor", 2);
  run<2>("v_mad_i32_i24", 1);
  run<3>("v_mul_lo_u32", 1);
  run<4>("v_bfe_u32 + v_add (sdwa?)", 2);
  run<5>("v_cmp + v_cndmask (+add)", 3);
  return 0;
}
