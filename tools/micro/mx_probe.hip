// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands: (1) which operand layout? (2) sustained rate under the power cap.
//   ./mx_probe layout A.bin B.bin    A: [16][128] e4m3 bytes, B: [16][128] e4m3 bytes (row j = column j of B, k-contiguous)
//       prints C[16][16] (fp32) for each layout hypothesis; tools/micro/mx_probe.py compares with the dequantised product
//   ./mx_probe rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// hypothesis h: byte q (0..31) of lane l holds k = kmap(h, l >> 4, q)
__device__ __host__ inline int kmap(int h, int g, int q) {
  switch (h) {
    case 0: return 32 * g + q;                                  // 32 consecutive k per lane group
    case 1: return 16 * g + (q & 15) + 64 * (q >> 4);           // two K = 64 halves, 16 consecutive k each
    case 2: return 8 * g + (q & 7) + 32 * (q >> 3);             // four K = 32 quarters, 8 consecutive k each
    default: return 4 * g + (q & 3) + 16 * (q >> 2);
  }
}

__global__ void layout_kernel(const uint8_t* A, const uint8_t* B, float* C, int h) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  union { i32x8 v; uint8_t b[32]; } a, b;
  for (int q = 0; q < 32; ++q) {
    a.b[q] = A[r * 128 + kmap(h, g, q)];
    b.b[q] = B[r * 128 + kmap(h, g, q)];
  }
  f32x4 acc = {0, 0, 0, 0};
  // operand order as in the bf16 GEMMs of this build: (b, a) so that a lane owns 4 consecutive output columns... here plain (a, b)
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, acc, 0, 0, 0, 127, 0, 127);
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];   // standard C/D map: col = lane & 15, row = 4 * (lane >> 4) + i
}

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* sink, int reps) {
  uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  float s = 0.f;
  if (MODE == 0) {   // MX fp8, 8 x 4 register tile of 16x16 blocks, K = 128 per instruction
    i32x8 av[8], bv[4];
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; av[q][i] = (int)(h & 0x3F3F3F3Fu); }   // |x| < 2
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; bv[q][i] = (int)(h & 0xBFBFBFBFu & 0x3F3F3F3Fu) | (int)(h & 0x80808080u); }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv[i & 3], av[i >> 2], acc[i], 0, 0, 0, 127, 0, 127);
    }
    for (int i = 0; i < 32; ++i) s += acc[i][0];
  } else {           // bf16 16x16x32, the same register tile (reference point, as tools/micro/mfma_power.hip)
    bf16x8 av[8], bv[4];
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; av[q][i] = (__bf16)(((int)(h >> 16) & 0xFFFF) / 65536.0f - 0.5f); }
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; bv[q][i] = (__bf16)(((int)(h >> 16) & 0xFFFF) / 65536.0f - 0.5f); }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[i & 3], av[i >> 2], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) s += acc[i][0];
  }
  if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void run_rate(const char* name, double macs_per_mfma, float* sink) {
  const int reps = 10000;
  rate_kernel<MODE><<<256, 512>>>(sink, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) rate_kernel<MODE><<<256, 512>>>(sink, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 20.0 * 256 * 8 * (double)reps * 32 * 2.0 * macs_per_mfma;
    printf("%s: %.1f ms -> %.0f TFLOP/s sustained\n", name, ms, flops / ms / 1e9);
  }
}

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "layout")) {
    uint8_t hA[2048], hB[2048];
    FILE* f = fopen(argv[2], "rb"); if (!f || fread(hA, 1, 2048, f) != 2048) return 2; fclose(f);
    f = fopen(argv[3], "rb"); if (!f || fread(hB, 1, 2048, f) != 2048) return 2; fclose(f);
    uint8_t *dA, *dB; float* dC; float hC[256];
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    for (int h = 0; h < 4; ++h) {
      layout_kernel<<<1, 64>>>(dA, dB, dC, h);
      hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
      printf("H%d", h);
      for (int i = 0; i < 256; ++i) printf(" %.6g", hC[i]);
      printf("\n");
    }
    return 0;
  }
  float* sink; hipMalloc(&sink, 4);
  run_rate<1>("v_mfma_f32_16x16x32_bf16, random operands", 16 * 16 * 32, sink);
  run_rate<0>("v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, unit scales), random operands", 16 * 16 * 128, sink);
  return 0;
}
