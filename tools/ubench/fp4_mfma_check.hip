// Does v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands compute the 0/1 inner products the all-pairs Hamming kernel needs?
// Lane (r = lane & 31, h = lane >> 5) supplies, for A, bits 32h .. 32h+31 of row r and, for B, the same bits of column r, one nibble per bit
// (0x2 = 1.0 in E2M1), nibble j of the 128-bit operand = bit 32h + j; scales E8M0 127 (= 1.0).  Any k order is fine as long as A and B use
// the same one.  Expected: C[row][col] = popcount(A_row & B_col), C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Also times 8 x i8 32x32x32 against 4 x fp4 32x32x64 (the same K = 256).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ inline uint32_t nib8(uint32_t b) {  // 8 bits -> 8 nibbles of 0x2
  const uint32_t lo = ((b & 0xfu) * 0x249u) & 0x1111u, hi = (((b >> 4) & 0xfu) * 0x249u) & 0x1111u;
  return (lo | (hi << 16)) << 1;
}
__global__ void k_check(const uint64_t* A, const uint64_t* B, float* C) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const uint32_t wa = (uint32_t)(A[r] >> (32 * h)), wb = (uint32_t)(B[r] >> (32 * h));
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
  for (int i = 0; i < 4; ++i) a[i] = (int)nib8(wa >> (8 * i)), b[i] = (int)nib8(wb >> (8 * i));
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}
template <int FP4>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
  v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {3, 2, 1, (int)threadIdx.x, 0, 0, 0, 0};
  v4i a4 = {1, 2, 3, 4}, b4 = {4, 3, 2, 1};
  v16f acc = {0};
  v16i acci = {0};
  for (int it = 0; it < iters; ++it) {
    if (FP4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, acci, 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + (float)acci[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  uint64_t hA[32], hB[32];
  srand(7);
  for (int i = 0; i < 32; ++i) hA[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand(), hB[i] = ((uint64_t)rand() << 35) ^ ((uint64_t)rand() << 13) ^ rand();
  uint64_t *dA, *dB;
  float* dC;
  (void)hipMalloc(&dA, sizeof hA), (void)hipMalloc(&dB, sizeof hB), (void)hipMalloc(&dC, 1024 * 4);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice), (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k_check<<<1, 64>>>(dA, dB, dC);
  float hC[1024];
  (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) bad += hC[i * 32 + j] != (float)__builtin_popcountll(hA[i] & hB[j]);
  printf("fp4 32x32x64 inner products of 0/1 rows: %d of 1024 wrong  (C[3][5] = %.1f, expected %d; C[5][3] = %.1f, expected %d)\n", bad, hC[3 * 32 + 5],
         __builtin_popcountll(hA[3] & hB[5]), hC[5 * 32 + 3], __builtin_popcountll(hA[5] & hB[3]));
  float* dO;
  (void)hipMalloc(&dO, 256 * 5 * 256 * 4);
  for (int fp4 = 0; fp4 < 2; ++fp4) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 20000;
    if (fp4) k_rate<1><<<1280, 256>>>(dO, 100); else k_rate<0><<<1280, 256>>>(dO, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    if (fp4) k_rate<1><<<1280, 256>>>(dO, iters); else k_rate<0><<<1280, 256>>>(dO, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f cycles per K = 256 step (32 x 32 tile) per SIMD at 5 wavefronts per SIMD, 2.4 GHz assumed\n", fp4 ? "4 x fp4 32x32x64" : "8 x i8 32x32x32 ", ms * 1e-3 * 2.4e9 / (5.0 * iters));
  }
  return 0;
}
