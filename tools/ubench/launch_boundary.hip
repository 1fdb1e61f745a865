// Micro-benchmark 5: what a dependent kernel boundary costs in one stream on this runtime (wall clock over a chain of launches,
// device time = chain / n once the host has run ahead), by kernel shape:
//   tiny          1 workgroup, no arguments to speak of
//   wide          2048 workgroups of 256 threads that exit at once
//   bigargs       the same with a 1 KB by-value argument struct (k_fast_score passes its level table that way)
//   scratch       a kernel whose code needs a private segment (a noinline callee with a local array: k_fast_score has one)
//   lds64         a kernel that declares 64 KB of LDS
//   +events       tiny, with hipEventRecord before and after every launch (what per-kernel profiling adds)
//   alternating   tiny and scratch kernels in turn (does the private-segment set-up cost once or at every change?)
// DESIGN.md section 7 (launch boundaries).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big {
  int v[256];
};
__global__ void k_tiny(int* p) {
  if (p && threadIdx.x == 999) *p = 1;
}
__global__ void k_big(int* p, Big b) {
  if (p && threadIdx.x == 999) *p = b.v[17];
}
__device__ __noinline__ int callee(int* p, int n) {
  volatile int loc[64];
  for (int i = 0; i < 64; ++i) loc[i] = i * n;
  return loc[n & 63] + (p ? *p : 0);
}
__global__ void k_scratch(int* p, int n) {
  if (n == 12345) *p = callee(p, n);
}
__global__ void k_lds(int* p) {
  __shared__ int s[16384];
  if (p && threadIdx.x == 999) s[threadIdx.x] = 1, *p = s[3];
}
template <class F>
static double chain(hipStream_t st, int n, F launch) {
  for (int i = 0; i < 200; ++i) launch(i);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  int* d;
  hipMalloc(&d, 64);
  Big b{};
  const int N = 20000;
  std::vector<hipEvent_t> ev(2);
  for (auto& e : ev) hipEventCreate(&e);
  printf("{\"unit\": \"us per launch, chain of %d dependent launches in one stream (host + device, whichever is slower)\",\n", N);
  printf(" \"tiny\": %.2f,\n", chain(st, N, [&](int) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d); }));
  printf(" \"wide\": %.2f,\n", chain(st, N, [&](int) { hipLaunchKernelGGL(k_tiny, dim3(2048), dim3(256), 0, st, d); }));
  printf(" \"bigargs\": %.2f,\n", chain(st, N, [&](int) { hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, st, d, b); }));
  printf(" \"scratch\": %.2f,\n", chain(st, N, [&](int) { hipLaunchKernelGGL(k_scratch, dim3(2048), dim3(256), 0, st, d, 1); }));
  printf(" \"lds64\": %.2f,\n", chain(st, N, [&](int) { hipLaunchKernelGGL(k_lds, dim3(2048), dim3(256), 0, st, d); }));
  printf(" \"tiny+events\": %.2f,\n", chain(st, N, [&](int) {
           hipEventRecord(ev[0], st);
           hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d);
           hipEventRecord(ev[1], st);
         }));
  printf(" \"alternating tiny/scratch\": %.2f\n}\n", chain(st, N, [&](int i) {
           if (i & 1)
             hipLaunchKernelGGL(k_scratch, dim3(2048), dim3(256), 0, st, d, 1);
           else
             hipLaunchKernelGGL(k_tiny, dim3(2048), dim3(256), 0, st, d);
         }));
  return 0;
}
