// Micro-benchmark 5: issue cost of the cross-lane and byte-shuffle instructions the FAST screen can be built from -- DPP moves that shift
// the whole wavefront by one lane (wave_shr / wave_shl, GFX9 only), the same shift folded into a VOP2 instruction, v_alignbyte,
// v_perm, v_lshl_or, the compare + mbcnt of a queue push -- and a check that the wavefront shift does what the screen needs
// (lane i reads lane i -+ 1 across the rows of 16).  Cycles per instruction per SIMD at 5 wavefronts per SIMD, four independent registers.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t a0) {
  uint32_t a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = a0 + threadIdx.x * (i + 1) * 2654435761u;
  uint32_t b = a0 * 77u + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      uint32_t& x = a[r % 4];
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
      if (OP == 1) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
      if (OP == 2) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
      if (OP == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
      if (OP == 4) asm volatile("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
      if (OP == 5) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(x) : "v"(b));
      if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a[(r + 1) % 4]));
      if (OP == 7) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(x) : "v"(b));
      if (OP == 8) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
      if (OP == 9) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(b));
      if (OP == 10) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(x));
      if (OP == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
      if (OP == 12) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a[(r + 1) % 4]));
      if (OP == 13) asm volatile("v_and_b32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
      if (OP == 14) asm volatile("v_or_b32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
      if (OP == 15) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
      if (OP == 16) asm volatile("v_bfe_u32 %0, %0, 1, 7" : "+v"(x));
      if (OP == 17) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
      if (OP == 18) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
      if (OP == 19) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a[(r + 1) % 4]));
      if (OP == 20) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
      if (OP == 21) asm volatile("v_not_b32 %0, %0" : "+v"(x));
      if (OP == 22) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a[(r + 1) % 4]));
      if (OP == 23) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a[(r + 1) % 4]));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name) {
  uint32_t* d;
  const int wps = 5, blocks = 256 * wps, iters = 4000;
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
  printf("%-44s %6.2f cycles per instruction per SIMD\n", name, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 32));
  (void)hipFree(d);
}
__global__ void k_check(uint32_t* out) {
  const uint32_t v = threadIdx.x + 100;
  uint32_t r = 7777, l = 8888;
  asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
  asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(l) : "v"(v));
  out[threadIdx.x] = r, out[64 + threadIdx.x] = l;
}
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 128 * 4);
  k_check<<<1, 64>>>(d);
  uint32_t h[128];
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int ok_r = h[0] == 7777, ok_l = h[127] == 8888;
  for (int i = 1; i < 64; ++i) ok_r &= h[i] == (uint32_t)(i - 1 + 100);
  for (int i = 0; i < 63; ++i) ok_l &= h[64 + i] == (uint32_t)(i + 1 + 100);
  for (int i = 0; i < 64; ++i) printf("%u%c", h[i], i % 16 == 15 ? '\n' : ' ');
  for (int i = 0; i < 64; ++i) printf("%u%c", h[64 + i], i % 16 == 15 ? '\n' : ' ');
  printf("wave_shr:1 gives lane i the value of lane i-1 (lane 0 keeps its own): %s   [lane0 %u lane16 %u lane32 %u]\n", ok_r ? "yes" : "NO", h[0], h[16], h[32]);
  printf("wave_shl:1 gives lane i the value of lane i+1 (lane 63 keeps its own): %s   [lane15 %u lane31 %u lane63 %u]\n", ok_l ? "yes" : "NO", h[64 + 15], h[64 + 31], h[127]);
  run<0>("v_add_u32");
  run<1>("v_mov_b32_dpp wave_shr:1");
  run<2>("v_mov_b32_dpp wave_shl:1");
  run<3>("v_mov_b32_dpp row_shr:1");
  run<15>("v_mov_b32_dpp quad_perm");
  run<4>("v_add_u32_dpp wave_shr:1");
  run<13>("v_and_b32_dpp wave_shl:1");
  run<14>("v_or_b32_dpp wave_shr:1");
  run<5>("v_alignbyte_b32");
  run<6>("v_perm_b32");
  run<7>("v_lshl_or_b32");
  run<12>("v_and_or_b32");
  run<22>("v_or3_b32");
  run<23>("v_add3_u32");
  run<19>("v_bfi_b32");
  run<16>("v_bfe_u32");
  run<8>("v_cmp_ne_u32 vcc");
  run<9>("v_mbcnt_lo_u32_b32");
  run<10>("v_lshrrev_b32");
  run<11>("v_and_b32");
  run<17>("v_sub_u32");
  run<18>("v_or_b32");
  run<20>("v_xor_b32");
  run<21>("v_not_b32");
  return 0;
}
