// Which MFMA shape sustains more FLOP/s under the power cap?  Register-only loops (no memory), 2 waves per SIMD, ~1 s each.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* sink, int reps) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.5f); }
  float s = 0;
  if (SHAPE == 17) {   // 16x16x32 with DIFFERENT random operands for consecutive MFMAs (switching activity of real data)
    bf16x8 av[8], bv[4];
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; av[q][i] = (__bf16)(((int)(h >> 16) & 0xFFFF) / 65536.0f - 0.5f); }
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; bv[q][i] = (__bf16)(((int)(h >> 16) & 0xFFFF) / 65536.0f - 0.5f); }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[i & 3], av[i >> 2], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) s += acc[i][0];
  } else if (SHAPE == 33) {   // 32x32x16, random operands, 4 x 2 register tile of 32x32 blocks, two k-steps (same MACs per pass as 17)
    bf16x8 av[4][2], bv[2][2];
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q) for (int t = 0; t < 2; ++t) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; av[q][t][i] = (__bf16)(((int)(h >> 16) & 0xFFFF) / 65536.0f - 0.5f); }
    for (int q = 0; q < 2; ++q) for (int t = 0; t < 2; ++t) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; bv[q][t][i] = (__bf16)(((int)(h >> 16) & 0xFFFF) / 65536.0f - 0.5f); }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[i & 1][t], av[i >> 1][t], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0];
  } else if (SHAPE == 16) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) s += acc[i][0];
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {       // 8 tiles x 2 k-steps of 16 = the same 32 x 8192 MACs as the loop above
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0];
  }
  if (s == 12345.678f) sink[0] = s;
}

template <int SHAPE>
static void run(const char* name, float* sink) {
  const int reps = 20000;
  k<SHAPE><<<256, 512>>>(sink, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<SHAPE><<<256, 512>>>(sink, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 20.0 * 256 * 8 * (double)reps * 32 * 2.0 * 8192;   // blocks x waves x reps x 32 MFMA-equivalents x 2 x MACs
    printf("%s: %.1f ms  -> %.0f TFLOP/s sustained\n", name, ms, flops / ms / 1e9);
  }
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  run<16>("v_mfma_f32_16x16x32_bf16", sink);
  run<32>("v_mfma_f32_32x32x16_bf16", sink);
  run<17>("v_mfma_f32_16x16x32_bf16, random operands, 8x4 register tile", sink);
  run<33>("v_mfma_f32_32x32x16_bf16, random operands, 4x2 register tile", sink);
  run<17>("v_mfma_f32_16x16x32_bf16, random operands, 8x4 register tile (again)", sink);
  return 0;
}
