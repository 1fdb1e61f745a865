// Micro-benchmark 3: issue cost per instruction type (inline asm, 8 independent registers per lane, 8 waves per SIMD) -- the
// numbers the FAST kernel's instruction mix is priced with.
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(ASM)                                                                                       \
  for (int it = 0; it < iters; ++it) {                                                                  \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                     \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));    \
    }                                                                                                   \
  }
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t a0) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x * (i + 1) * 2654435761u;
  uint32_t b = a0 * 77u + threadIdx.x, c = a0 * 13u + 5u;
  if (OP == 0) BODY("v_add_u32 %0, %0, %1")
  if (OP == 1) BODY("v_min_i32 %0, %0, %1")
  if (OP == 2) BODY("v_min3_i32 %0, %0, %1, %2")
  if (OP == 3) BODY("v_pk_min_u16 %0, %0, %1")
  if (OP == 4) BODY("v_pk_min_i16 %0, %0, %1")
  if (OP == 5) BODY("v_pk_sub_u16 %0, %0, %1 clamp")
  if (OP == 6) BODY("v_and_b32 %0, %0, %1")
  if (OP == 7) BODY("v_alignbyte_b32 %0, %0, %1, %2")
  if (OP == 8) BODY("v_perm_b32 %0, %0, %1, %2")
  if (OP == 9) BODY("v_mbcnt_lo_u32_b32 %0, %1, %0")
  if (OP == 10) BODY("v_sub_u32 %0, %0, %1")
  if (OP == 11) BODY("v_lshl_add_u32 %0, %0, 2, %1")
  if (OP == 12) BODY("v_max3_i32 %0, %0, %1, %2")
  if (OP == 13) BODY("v_med3_i32 %0, %0, %1, %2")
  if (OP == 14) BODY("v_min_u16 %0, %0, %1")
  if (OP == 15) BODY("v_sad_u8 %0, %0, %1, %2")
  if (OP == 16) BODY("v_bfe_u32 %0, %0, 8, 8")
  if (OP == 17) BODY("v_and_or_b32 %0, %0, %1, %2")
  if (OP == 18) BODY("v_add3_u32 %0, %0, %1, %2")
  if (OP == 19) BODY("v_mov_b32 %0, %1")
  if (OP == 20) BODY("v_cmp_gt_u32 vcc, %0, %1")
  if (OP == 21) BODY("v_min_i32 %0, %0, %1\n v_max_i32 %0, %0, %2")
  if (OP == 22) BODY("v_pk_max_i16 %0, %0, %1\n v_pk_min_i16 %0, %0, %2")
  if (OP == 23) BODY("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
  if (OP == 24) BODY("v_min_i16 %0, %0, %1")
  if (OP == 25) BODY("v_msad_u8 %0, %0, %1, %2")
  if (OP == 26) BODY("v_min_u32 %0, %0, %1")
  if (OP == 27) BODY("v_xor_b32 %0, %0, %1")
  if (OP == 28) BODY("v_lshlrev_b32 %0, 3, %0")
  if (OP == 29) BODY("v_lshrrev_b32 %0, %1, %0")
  if (OP == 30) BODY("v_or_b32 %0, %0, %1")
  if (OP == 31) BODY("v_mul_u32_u24 %0, %0, %1")
  if (OP == 32) BODY("v_cndmask_b32 %0, %0, %1, vcc")
  if (OP == 33) BODY("v_sub_u16 %0, %0, %1")
  if (OP == 34) BODY("v_max_u16 %0, %0, %1")
  if (OP == 35) BODY("v_add_f32 %0, %0, %1")
  if (OP == 36) BODY("v_fma_f32 %0, %0, %1, %2")
  if (OP == 37) BODY("v_min_f32 %0, %0, %1")
  if (OP == 38) BODY("v_cvt_f32_ubyte0 %0, %0")
  if (OP == 39) BODY("v_dot4_u32_u8 %0, %0, %1, %2")
  if (OP == 40) BODY("v_mul_lo_u32 %0, %0, %1")
  if (OP == 41) BODY("v_mul_hi_u32 %0, %0, %1")
  if (OP == 42) BODY("v_mad_u32_u24 %0, %0, %1, %2")
  if (OP == 43) BODY("v_mul_f32 %0, %0, %1")
  if (OP == 44) BODY("v_fmac_f32 %0, %1, %2")
  if (OP == 45) BODY("v_ashrrev_i32 %0, 1, %0")
  if (OP == 46) BODY("v_mad_u16 %0, %0, %1, %2")
  if (OP == 47) BODY("v_mul_lo_u16 %0, %0, %1")
  if (OP == 48) BODY("v_pk_mul_lo_u16 %0, %0, %1")
  if (OP == 49) BODY("v_pk_mad_u16 %0, %0, %1, %2")
  if (OP == 50) BODY("v_cvt_f32_u32 %0, %0")
  if (OP == 51) BODY("v_add_u16 %0, %0, %1")
  if (OP == 52) BODY("v_max_i16 %0, %0, %1")
    if (OP == 54) BODY("v_addc_co_u32 %0, vcc, %0, %1, vcc")
  if (OP == 55) BODY("v_add_co_u32 %0, vcc, %0, %1")
  if (OP == 60) BODY("v_cndmask_b32 %0, %0, %1, s[20:21]")
  if (OP == 61) BODY("v_cmp_gt_u32 s[20:21], %0, %1")
  if (OP == 62) BODY("v_bfi_b32 %0, %0, %1, %2")
  if (OP == 63) BODY("v_lshl_or_b32 %0, %0, 8, %1")
  if (OP == 64) BODY("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c")
  if (OP == 65) BODY("v_pk_add_u16 %0, %0, %1")
  if (OP == 66) BODY("v_floor_f32 %0, %0")
  if (OP == 67) BODY("v_rndne_f32 %0, %0")
  if (OP == 68) BODY("v_cvt_u32_f32 %0, %0")
  if (OP == 71) BODY("v_max_f32 %0, %0, %1")
  if (OP == 72) BODY("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
  if (OP == 73) BODY("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
  if (OP == 74) BODY("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
  if (OP == 75) BODY("v_cvt_f32_ubyte2 %0, %0")
  if (OP == 76) BODY("v_subrev_u32 %0, %0, %1")
  if (OP == 77) BODY("v_mad_u32_u16 %0, %0, %1, %2")
  if (OP == 78) BODY("v_add_u32 %0, 5, %0")
  if (OP == 79) BODY("v_add_u32 %0, s20, %0")
  if (OP == 80) BODY("v_add_u32 %0, 0x12345, %0")
  if (OP == 81) BODY("v_sub_f32 %0, %0, %1")
  if (OP == 82) BODY("v_mul_u32_u24 %0, 3, %0")
  if (OP == 90 || OP == 91 || OP == 92) {   // 64-bit address arithmetic
    uint64_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = ((uint64_t)a[i] << 32) | a[i + 4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (OP == 90) asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c));
          if (OP == 91) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c));
          if (OP == 92) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 3]));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int per = 1) {
  uint32_t* d;
  const int blocks = 256 * 8, iters = 4000;
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
  const double wi = (double)blocks * 4 * iters * 32 * per;
  printf("%-28s %8.3f ms  %7.1f wave-instr/ns = %5.2f cycles/instr/SIMD @2.4GHz\n", name, ms, wi / (ms * 1e6), 1024.0 * 2.4 / (wi / (ms * 1e6)));
  (void)hipFree(d);
}
int main() {
  run<0>("v_add_u32"); run<1>("v_min_i32"); run<26>("v_min_u32"); run<27>("v_xor_b32"); run<2>("v_min3_i32"); run<12>("v_max3_i32"); run<13>("v_med3_i32");
  run<3>("v_pk_min_u16"); run<4>("v_pk_min_i16"); run<5>("v_pk_sub_u16 clamp"); run<14>("v_min_u16"); run<24>("v_min_i16");
  run<6>("v_and_b32"); run<17>("v_and_or_b32"); run<18>("v_add3_u32"); run<11>("v_lshl_add_u32");
  run<7>("v_alignbyte_b32"); run<8>("v_perm_b32"); run<15>("v_sad_u8"); run<25>("v_msad_u8"); run<16>("v_bfe_u32"); run<9>("v_mbcnt_lo");
  run<10>("v_sub_u32"); run<23>("v_sub_u32_sdwa"); run<19>("v_mov_b32"); run<20>("v_cmp_gt_u32");
  run<28>("v_lshlrev_b32"); run<29>("v_lshrrev_b32"); run<45>("v_ashrrev_i32"); run<30>("v_or_b32"); run<31>("v_mul_u32_u24"); run<42>("v_mad_u32_u24"); run<32>("v_cndmask_b32");
  run<51>("v_add_u16"); run<33>("v_sub_u16"); run<34>("v_max_u16"); run<52>("v_max_i16"); run<46>("v_mad_u16"); run<47>("v_mul_lo_u16"); run<48>("v_pk_mul_lo_u16"); run<49>("v_pk_mad_u16");
  run<35>("v_add_f32"); run<43>("v_mul_f32"); run<36>("v_fma_f32"); run<44>("v_fmac_f32"); run<37>("v_min_f32"); run<38>("v_cvt_f32_ubyte0"); run<50>("v_cvt_f32_u32");
  run<39>("v_dot4_u32_u8"); run<40>("v_mul_lo_u32"); run<41>("v_mul_hi_u32"); run<55>("v_add_co_u32"); run<54>("v_addc_co_u32");
  run<60>("v_cndmask_b32 (sgpr mask)"); run<61>("v_cmp_gt_u32 -> sgpr"); run<62>("v_bfi_b32"); run<63>("v_lshl_or_b32"); run<64>("v_bitop3_b32"); run<65>("v_pk_add_u16");
  run<66>("v_floor_f32"); run<67>("v_rndne_f32"); run<68>("v_cvt_u32_f32"); run<71>("v_max_f32"); run<81>("v_sub_f32");
  run<72>("v_mov_b32_dpp row_shr:1"); run<73>("v_mov_b32_dpp wave_shr:1"); run<74>("v_add_u32_dpp row_shr:1"); run<75>("v_cvt_f32_ubyte2"); run<76>("v_subrev_u32"); run<77>("v_mad_u32_u16");
  run<78>("v_add_u32 inline const"); run<79>("v_add_u32 sgpr"); run<80>("v_add_u32 literal"); run<82>("v_mul_u32_u24 const");
  run<90>("v_mad_i64_i32"); run<91>("v_mad_u64_u32"); run<92>("v_lshl_add_u64");
  run<21>("min_i32+max_i32", 2); run<22>("pk_max_i16+pk_min_i16", 2);
  return 0;
}
