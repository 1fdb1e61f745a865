// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses.
// MI355X_MICROARCH.md: FETCH_SIZE reports exactly half the bytes of a 16-B-per-lane coalesced streaming read; "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern".  k_fast_score / k_gauss7 load one dword per lane,
// k_pad_level0 16 bytes, k_describe short row segments.  Each kernel below streams a buffer of known size once (larger than the
// 256 MiB Infinity Cache, so nothing is served on-die) with one load width; run under
//   rocprofv3 --pmc FETCH_SIZE -- tools/ubench/stream_read       (and --pmc WRITE_SIZE in a second pass)
// and divide the known bytes by the counter (tools/fetch_calibration.py does it).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_read tools/ubench/stream_read.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <class T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ src, size_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = src[i];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    if constexpr (sizeof(T) >= 4) {
#pragma unroll
      for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
    } else {
      acc ^= (uint32_t)v;
    }
  }
  if (acc == 77u) sink[0] = acc;  // never true in practice (the buffer holds ones): keeps the loads alive without a store stream
}
// rows of `seg` bytes every `pitch` bytes, one lane per 4 bytes of a segment (k_describe's window rows: 40 of 704 bytes)
__global__ __launch_bounds__(256) void k_read_segments(const uint8_t* __restrict__ src, size_t rows, int pitch, int seg_dwords, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const size_t total = rows * (size_t)seg_dwords;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / seg_dwords, d = i - r * seg_dwords;
    acc ^= *reinterpret_cast<const uint32_t*>(src + r * (size_t)pitch + 4 * d);
  }
  if (acc == 77u) sink[0] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    if constexpr (sizeof(T) >= 4) {
#pragma unroll
      for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = (uint32_t)i + k;
    } else {
      v = (T)i;
    }
    dst[i] = v;
  }
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: four times the Infinity Cache
  uint8_t* buf;
  uint32_t* sink;
  if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 256) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  const int grid = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_read<uint8_t>, dim3(grid), dim3(256), 0, 0, buf, bytes, sink);
    hipLaunchKernelGGL(k_read<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_read<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(k_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(k_read_segments, dim3(grid), dim3(256), 0, 0, buf, bytes / 704, 704, 10, sink);
    hipLaunchKernelGGL(k_write<uint8_t>, dim3(grid), dim3(256), 0, 0, buf, bytes);
    hipLaunchKernelGGL(k_write<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4);
    hipLaunchKernelGGL(k_write<uint4>, dim3(grid), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("{\"bytes\": %zu, \"segment_bytes\": %zu, \"segment_footprint_bytes\": %zu}\n", bytes, (bytes / 704) * 40, bytes);
  return 0;
}
