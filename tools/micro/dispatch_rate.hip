// How fast does the hardware launch workgroups?  Empty kernels (one LDS touch), varying workgroup shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* sink) {
  extern __shared__ int lds[];
  if (threadIdx.x == 0) lds[0] = blockIdx.x;
  __syncthreads();
  if (lds[0] == -1) sink[0] = 1;
}
static void run(int blocks, int threads, int lds_bytes, int* sink) {
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  for (int i = 0; i < 3; ++i) k<<<blocks, threads, lds_bytes>>>(sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int it = 50;
  for (int i = 0; i < it; ++i) k<<<blocks, threads, lds_bytes>>>(sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%5d workgroups x %3d threads, %3d KiB LDS: %7.1f us per launch -> %.2f workgroups/us (%.3f us per workgroup and XCD)\n",
         blocks, threads, lds_bytes / 1024, ms / it * 1e3, blocks / (ms / it * 1e3), (ms / it * 1e3) / (blocks / 8.0));
}
int main() {
  int* sink; hipMalloc(&sink, 4);
  for (int b : {256, 1536, 6144}) run(b, 512, 128 * 1024, sink);
  for (int b : {512, 1536, 6144}) run(b, 256, 64 * 1024, sink);
  for (int b : {1536, 6144, 24576}) run(b, 256, 1024, sink);
  return 0;
}
