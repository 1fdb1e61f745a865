// Probe: how much dynamic LDS a launch may ask for on this device, with and without hipFuncAttributeMaxDynamicSharedMemorySize; and a static 160 KiB kernel.
//   hipcc --offload-arch=gfx950 -O2 -o lds_limit lds_limit.hip && ./lds_limit
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T>
__global__ __launch_bounds__(T) void k_dyn(int* out, int n) {
  extern __shared__ int sm[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sm[n - 1];
}
__global__ __launch_bounds__(1024) void k_static(int* out) {
  __shared__ int sm[160 * 1024 / 4];
  for (int i = threadIdx.x; i < 160 * 256; i += blockDim.x) sm[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sm[160 * 256 - 1];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
  int* d;
  hipMalloc(&d, 4096);
  for (int kb : {32, 64, 65, 96, 128, 160}) {
    for (int attr = 0; attr < 2; ++attr) {
      hipError_t ea = hipSuccess;
      if (attr) ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dyn<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
      hipLaunchKernelGGL(k_dyn<1024>, dim3(8), dim3(1024), kb * 1024, 0, d, kb * 256);
      hipError_t el = hipGetLastError();
      hipError_t es = hipDeviceSynchronize();
      int v = -1;
      hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost);
      printf("dyn %3d KiB threads 1024 attr %d: setattr %d launch %d (%s) sync %d value %d\n", kb, attr, (int)ea, (int)el, hipGetErrorString(el), (int)es, v);
    }
  }
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_dyn<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int kb : {96, 160}) {
    hipLaunchKernelGGL(k_dyn<256>, dim3(8), dim3(256), kb * 1024, 0, d, kb * 256);
    hipError_t el = hipGetLastError();
    hipDeviceSynchronize();
    printf("dyn %3d KiB threads 256 attr 160K: launch %d (%s)\n", kb, (int)el, hipGetErrorString(el));
  }
  hipLaunchKernelGGL(k_static, dim3(8), dim3(1024), 0, 0, d);
  hipError_t el = hipGetLastError();
  hipError_t es = hipDeviceSynchronize();
  int v = -1;
  hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost);
  printf("static 160 KiB: launch %d (%s) sync %d value %d\n", (int)el, hipGetErrorString(el), (int)es, v);
  return 0;
}
