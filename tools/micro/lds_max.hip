// largest dynamic LDS allocation a kernel can be launched with on this GPU
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int n) { extern __shared__ int s[]; s[threadIdx.x] = n; __syncthreads(); if (threadIdx.x == 0) out[0] = s[63] + n; }
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu  sharedMemPerBlockOptin %zu  maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.maxSharedMemoryPerMultiProcessor);
  int* d; (void)hipMalloc(&d, 4);
  for (int kb : {64, 128, 152, 159, 160}) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    k<<<1, 512, kb * 1024>>>(d, kb);
    hipError_t e2 = hipDeviceSynchronize();
    printf("%d KB: setattr %s, launch %s\n", kb, hipGetErrorString(e), hipGetErrorString(e2 == hipSuccess ? hipGetLastError() : e2));
  }
  return 0;
}
