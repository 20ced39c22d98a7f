// calibration: s_memtime ticks per back-to-back v_mfma_f32_16x16x32_bf16 (32 independent accumulators), one or two waves per SIMD
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void k(uint64_t* out, float* sink, int reps) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  uint64_t* d; float* sink;
  hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4);
  for (int threads : {256, 512}) {
    const int reps = 200;
    k<<<256, threads>>>(d, sink, reps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<<<256, threads>>>(d, sink, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (reps * 32.0);
    printf("%d waves/CU (%d per SIMD): %.2f s_memtime ticks per MFMA per wave; kernel %.3f ms -> %.1f ns per MFMA per wave -> tick = %.3f ns\n",
           threads / 64, threads / 256, per, ms, ms * 1e6 / (reps * 32.0), ms * 1e6 / (reps * 32.0) / per);
  }
  return 0;
}
