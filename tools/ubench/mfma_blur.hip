// Micro-benchmark / prototype: the separable 7 x 7 blur of src/ORBextractor.cc:942 on the int8 matrix cores (v_mfma_i32_32x32x32_i8) -- the gate
// for moving the production blur (csrc/gauss_body.hpp) off the vector ALU.  Interior pixels only, row-major planes, no borders, no tiled output:
// an upper bound for what a production form could reach, checked bit for bit against a CPU evaluation of the exact integer sums.
//
//   row pass   : S[r][x] = sum_k tap[k] p[r][x + k - 3]           as (32 rows x 32 input columns) x (32 x 32 banded Toeplitz), two MFMAs per
//                32 x 32 outputs (input columns [X - 16, X + 16) and [X + 16, X + 48)); pixels biased by -128 (xor 0x80), the accumulator
//                starts at 128 * sum(taps), so it holds S itself (16 bits)
//   column pass: out[y][x] = sum_k tap[k] S[y + k - 3][x]         S split into hi / lo bytes (xor 0x80 each), the packed row-pass result of a
//                lane IS an A operand (m = column, k = rows in the order the accumulator holds them); B = the Toeplitz in that row order;
//                C'[x][y]: a lane ends up with four consecutive x of one row per register group -- dwords ready to store; output rows are
//                shifted by 16 against the input row blocks, so every output block takes two S blocks: four MFMAs (hi, lo) x (this, next)
//   rounding   : sum / 65536 to nearest even + clamp = v_cvt_pk_u8_f32 of the exact float (the x86-64 contract, csrc/gauss_body.hpp)
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_blur mfma_blur.hip && ./mfma_blur
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int tap_of(int idx) {
  const int t[7] = {18, 34, 49, 55, 49, 34, 18};
  return idx >= 0 && idx <= 6 ? t[idx] : 0;
}
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) { return (uint32_t)(a & 255) | (uint32_t)(b & 255) << 8 | (uint32_t)(c & 255) << 16 | (uint32_t)(d & 255) << 24; }

// wave = a strip of 32 output columns [X, X + 32), walking `nblk` output row blocks of 32 rows from row Y0 (a multiple of 32) + 16 on
// probe: 0 = the kernel; 1 = without its stores (kept only for an impossible value); 2 = every block reads the plane's first 32 rows (cache-resident)
template <int PROBE>
__global__ __launch_bounds__(256) void k_mfma_blur(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int pitch, int h, int strips_x, int nblk) {
  const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int sx = wave % strips_x, seg = wave / strips_x;
  const int X = 16 + sx * 32;          // first output column (columns [16, pitch - 48) are covered)
  const int b0 = seg * nblk;           // first input row block
  // constant operands
  v4i B1, B2, T1, T2;
  {
    int b1[4], b2[4], t1[4], t2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int q1[4], q2[4], r1[4], r2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * hh + 4 * i + j;      // B operand of the row pass: k = input column of the window
        q1[j] = tap_of(k - n - 13), q2[j] = tap_of(k - n + 19);
        const int ro = 8 * i + 4 * hh + j;      // B operand of the column pass: k runs over the rows in accumulator order
        r1[j] = tap_of(ro - n - 13), r2[j] = tap_of(ro - n + 19);
      }
      b1[i] = (int)pack4(q1[0], q1[1], q1[2], q1[3]), b2[i] = (int)pack4(q2[0], q2[1], q2[2], q2[3]);
      t1[i] = (int)pack4(r1[0], r1[1], r1[2], r1[3]), t2[i] = (int)pack4(r2[0], r2[1], r2[2], r2[3]);
    }
    B1 = v4i{b1[0], b1[1], b1[2], b1[3]}, B2 = v4i{b2[0], b2[1], b2[2], b2[3]};
    T1 = v4i{t1[0], t1[1], t1[2], t1[3]}, T2 = v4i{t2[0], t2[1], t2[2], t2[3]};
  }
  v16i zero16, rinit, linit;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0, rinit[i] = 128 * 257, linit[i] = 257 * (32768 + 128);
  v16i accH = zero16, accL = linit;   // the output block in flight (rows 32 (b - 1) + 16 ..)
  auto load_block = [&](int b, v4i& A1, v4i& A2) {
    const int r = PROBE == 2 ? n : 32 * b + n;
    const uint8_t* p = src + (int64_t)(r < h ? r : h - 1) * pitch + X - 16 + 16 * hh;
    A1 = *reinterpret_cast<const v4i*>(p), A2 = *reinterpret_cast<const v4i*>(p + 32);
  };
  v4i N1, N2;
  load_block(b0, N1, N2);
  for (int b = b0; b <= b0 + nblk; ++b) {
    // row pass of input row block b; the next block's pixels are fetched meanwhile (a wavefront walks its blocks one after another)
    v4i A1 = N1, A2 = N2;
    load_block(b + 1, N1, N2);
#pragma unroll
    for (int i = 0; i < 4; ++i) A1[i] ^= (int)0x80808080, A2[i] ^= (int)0x80808080;
    v16i s = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B1, rinit, 0, 0, 0);
    s = __builtin_amdgcn_mfma_i32_32x32x32_i8(A2, B2, s, 0, 0, 0);
    // S (16 bits) -> hi / lo byte operands: register g of the operand = the bytes of accumulators 4g .. 4g + 3
    v4i Ahi, Alo;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint32_t a0 = (uint32_t)s[4 * g], a1 = (uint32_t)s[4 * g + 1], a2 = (uint32_t)s[4 * g + 2], a3 = (uint32_t)s[4 * g + 3];
      const uint32_t p01 = __builtin_amdgcn_perm(a1, a0, 0x05040100u);  // (lo0, hi0, lo1, hi1)
      const uint32_t p23 = __builtin_amdgcn_perm(a3, a2, 0x05040100u);
      Alo[g] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
      Ahi[g] = (int)(__builtin_amdgcn_perm(p23, p01, 0x07050301u) ^ 0x80808080u);
    }
    if (b > b0) {  // finish output block b - 1 with this block as its lower half, store it
      accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ahi, T2, accH, 0, 0, 0);
      accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(Alo, T2, accL, 0, 0, 0);
      const int y = 32 * (b - 1) + 16 + n;
      if (y < h - 16) {
        uint8_t* o = dst + (int64_t)y * pitch + X + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint32_t out = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t S = ((uint32_t)accH[4 * g + e] << 8) + (uint32_t)accL[4 * g + e];
            out = __builtin_amdgcn_cvt_pk_u8_f32((float)S * (1.0f / 65536.0f), (uint32_t)e, out);
          }
          if (PROBE != 1 || out == 0x12345678u) *reinterpret_cast<uint32_t*>(o + 8 * g) = out;
        }
      }
    }
    // start output block b with this block as its upper half
    accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ahi, T1, zero16, 0, 0, 0);
    accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(Alo, T1, linit, 0, 0, 0);
  }
}

int main() {
  const int taps[7] = {18, 34, 49, 55, 49, 34, 18};
  // ---- correctness on a small plane ----
  {
    const int pitch = 256, h = 224;
    std::vector<uint8_t> img((size_t)pitch * h), ref((size_t)pitch * h, 0), out((size_t)pitch * h, 0);
    srand(7);
    for (auto& v : img) v = (uint8_t)(rand() % 5 == 0 ? 255 : rand() & 255);
    for (int y = 0; y < 40; ++y)
      for (int x = 0; x < pitch; ++x) img[(size_t)y * pitch + x] = 255;  // a saturating band
    for (int y = 16; y < h - 16; ++y)
      for (int x = 16; x < pitch - 48; ++x) {
        long long S = 0;
        for (int j = 0; j < 7; ++j) {
          long long rs = 0;
          for (int i = 0; i < 7; ++i) rs += taps[i] * img[(size_t)(y + j - 3) * pitch + x + i - 3];
          S += taps[j] * rs;
        }
        long long q = S >> 16, rem = S & 0xffff;
        if (rem > 0x8000 || (rem == 0x8000 && (q & 1))) ++q;  // to nearest even
        ref[(size_t)y * pitch + x] = (uint8_t)(q > 255 ? 255 : q);
      }
    uint8_t *ds, *dd;
    hipMalloc(&ds, img.size()), hipMalloc(&dd, img.size());
    hipMemcpy(ds, img.data(), img.size(), hipMemcpyHostToDevice);
    hipMemset(dd, 0, img.size());
    const int strips_x = (pitch - 64) / 32, nblk = 2, segs = (h / 32 + nblk - 1) / nblk;
    k_mfma_blur<0><<<(strips_x * segs + 3) / 4, 256>>>(ds, dd, pitch, h, strips_x, nblk);
    hipMemcpy(out.data(), dd, out.size(), hipMemcpyDeviceToHost);
    long bad = 0, cnt = 0;
    for (int y = 16; y < h - 16; ++y)
      for (int x = 16; x < 16 + strips_x * 32; ++x) {
        ++cnt;
        if (out[(size_t)y * pitch + x] != ref[(size_t)y * pitch + x] && ++bad <= 5) printf("mismatch (%d, %d): %d vs %d\n", x, y, out[(size_t)y * pitch + x], ref[(size_t)y * pitch + x]);
      }
    printf("correctness: %ld of %ld pixels differ (launch %s)\n", bad, cnt, hipGetErrorString(hipGetLastError()));
    hipFree(ds), hipFree(dd);
  }
  // ---- speed: as many pixels as one bench step blurs (257 frames x 1 014 311 px = 260.7 M) ----
  {
    const int pitch = 4096, h = 65536;
    uint8_t *ds, *dd;
    hipMalloc(&ds, (size_t)pitch * h), hipMalloc(&dd, (size_t)pitch * h);
    hipMemset(ds, 77, (size_t)pitch * h);
    const int strips_x = (pitch - 64) / 32;
    auto run = [&](int probe, int nblk) {
      const int segs = (h / 32 + nblk - 1) / nblk;
      const int blocks = (strips_x * segs + 3) / 4;
      hipEvent_t a, b;
      hipEventCreate(&a), hipEventCreate(&b);
      auto go = [&]() {
        if (probe == 0) k_mfma_blur<0><<<blocks, 256>>>(ds, dd, pitch, h, strips_x, nblk);
        if (probe == 1) k_mfma_blur<1><<<blocks, 256>>>(ds, dd, pitch, h, strips_x, nblk);
        if (probe == 2) k_mfma_blur<2><<<blocks, 256>>>(ds, dd, pitch, h, strips_x, nblk);
      };
      go();
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int i = 0; i < 5; ++i) go();
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double px = (double)strips_x * 32 * h;
      printf("%-28s rows per wavefront %4d: %.3f ms per 260.7 M pixels (the production blur alone: ~0.17 ms), %.0f GB/s read + write\n",
             probe == 0 ? "kernel" : (probe == 1 ? "without its stores" : "reads from 32 resident rows"), nblk * 32, ms / 5 * 260.7e6 / px, 2 * px / (ms / 5 * 1e-3) / 1e9);
    };
    for (int nblk : {2, 4, 8, 16}) run(0, nblk);
    run(1, 16), run(2, 16);
  }
  return 0;
}
