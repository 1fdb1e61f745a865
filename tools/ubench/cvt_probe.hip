// what v_cvt_pk_u8_f32 does with fractions and out-of-range values (rounding mode, saturation)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, uint32_t* out, int n) {
  int i = threadIdx.x;
  if (i < n % 1000) {
    uint32_t r = 0xAABBCCDDu;
    if (n > 100) __builtin_amdgcn_s_setreg(0x801, 3);  // FP32 round mode: toward zero
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(in[i]));
    out[i] = r;
  }
}
int main(int argc, char**) {
  float h[] = {0.4f, 0.5f, 0.6f, 0.999f, 1.5f, 2.5f, 3.5f, 254.5f, 254.999f, 255.5f, 256.7f, 300.f, -3.f, 16777216.f, 1.0f, 2.0f};
  const int n = sizeof(h) / 4;
  float* di; uint32_t* dout; uint32_t o[32];
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, 4 * n);
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout, n + (argc > 1 ? 1000 : 0));
  hipMemcpy(o, dout, 4 * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%12.4f -> %08x (byte1 = %u)\n", h[i], o[i], (o[i] >> 8) & 0xff);
  return 0;
}
