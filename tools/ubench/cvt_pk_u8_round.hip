// Probe: how v_cvt_pk_u8_f32 rounds under the wavefront's fp32 rounding mode (default: nearest even; MODE.fp_round = 3: toward zero).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, uint32_t* out, int n, int rtz) {
  if (rtz) __builtin_amdgcn_s_setreg(0x801, 3);
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}
int main() {
  const float v[] = {0.5f, 1.5f, 2.5f, 3.5f, 2.7f, 3.49f, 254.5f, 255.5f, 255.49f, 256.0f, 300.f, -0.5f, -1.f, 0.49999997f, 1.5000001f, 2.4999998f, 126.5f, 127.5f};
  const int n = sizeof(v) / sizeof(v[0]);
  float* d;
  uint32_t* o;
  hipMalloc(&d, sizeof(v)), hipMalloc(&o, n * 4);
  hipMemcpy(d, v, sizeof(v), hipMemcpyHostToDevice);
  for (int rtz = 0; rtz < 2; ++rtz) {
    k<<<1, 64>>>(d, o, n, rtz);
    uint32_t r[64];
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    printf("mode %s:", rtz ? "toward zero" : "default");
    for (int i = 0; i < n; ++i) printf(" %g->%u", v[i], r[i]);
    printf("\n");
  }
  return 0;
}
