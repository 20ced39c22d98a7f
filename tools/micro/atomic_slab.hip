// What does it cost to ADD split-K partial tiles into the fp32 gradient with float atomics instead of writing slabs and
// reducing them in a second launch?  Shape of the FFN weight gradient: 3072 x 768 fp32 = 36 tiles of 256 x 256, 7 K-splits,
// 252 workgroups of 512 threads.  Build: hipcc --offload-arch=gfx950 -O3 -o atomic_slab atomic_slab.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int M = 3072, N = 768, TM = 256, TN = 256, TILES = (M / TM) * (N / TN), SPLITS = 7;

__global__ __launch_bounds__(512) void slab_store(float* __restrict__ slab, float v) {
  const int t = blockIdx.x % TILES, z = blockIdx.x / TILES, tm = t / (N / TN), tn = t % (N / TN);
  float* base = slab + (size_t)z * M * N + (size_t)tm * TM * N + tn * TN;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = w; r < TM; r += 8) *(f32x4*)(base + (size_t)r * N + lane * 4) = f32x4{v, v, v, v};
}
__global__ __launch_bounds__(256) void slab_reduce(const float* __restrict__ slab, float* __restrict__ C) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < (size_t)M * N; i += (size_t)gridDim.x * 1024) {
    f32x4 s = *(const f32x4*)(slab + i);
    for (int z = 1; z < SPLITS; ++z) s += *(const f32x4*)(slab + (size_t)z * M * N + i);
    *(f32x4*)(C + i) = s;
  }
}
// lanes = consecutive columns (256 contiguous bytes per wave instruction)
__global__ __launch_bounds__(512) void atomic_rows(float* __restrict__ C, float v) {
  const int t = blockIdx.x % TILES, tm = t / (N / TN), tn = t % (N / TN);
  float* base = C + (size_t)tm * TM * N + tn * TN;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = w; r < TM; r += 8)
    for (int c = lane; c < TN; c += 64) atomicAdd(base + (size_t)r * N + c, v);
}
// the accumulator layout of an un-swapped 16x16 MFMA: 16 lanes = 16 consecutive columns (64 B), 4 row groups per instruction
__global__ __launch_bounds__(512) void atomic_mfma(float* __restrict__ C, float v) {
  const int t = blockIdx.x % TILES, tm = t / (N / TN), tn = t % (N / TN);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = w >> 1, wn = w & 1;   // 4 x 2 waves of 64 x 128
  float* base = C + (size_t)(tm * TM + wm * 64) * N + tn * TN + wn * 128;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j)
      for (int e = 0; e < 4; ++e) atomicAdd(base + (size_t)(16 * i + 4 * (lane >> 4) + e) * N + 16 * j + (lane & 15), v);
}

int main() {
  float *slab, *C;
  hipMalloc(&slab, (size_t)SPLITS * M * N * 4);
  hipMalloc(&C, (size_t)M * N * 4);
  hipMemset(C, 0, (size_t)M * N * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.2f us\n", name, ms * 1000 / 50);
  };
  timeit("slab stores (7 x 9.4 MB)", [&] { slab_store<<<TILES * SPLITS, 512>>>(slab, 1.f); });
  timeit("slab reduce", [&] { slab_reduce<<<1152, 256>>>(slab, C); });
  timeit("stores + reduce", [&] { slab_store<<<TILES * SPLITS, 512>>>(slab, 1.f); slab_reduce<<<1152, 256>>>(slab, C); });
  hipMemset(C, 0, (size_t)M * N * 4);
  timeit("atomics, 256 B rows", [&] { atomic_rows<<<TILES * SPLITS, 512>>>(C, 1.f); });
  std::vector<float> h((size_t)M * N);
  hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
  double bad = 0; for (float x : h) bad += (x != 53.f * SPLITS);
  printf("  atomic_rows check: %g wrong of %zu (expect %g)\n", bad, h.size(), 53.0 * SPLITS);
  hipMemset(C, 0, (size_t)M * N * 4);
  timeit("atomics, MFMA layout", [&] { atomic_mfma<<<TILES * SPLITS, 512>>>(C, 1.f); });
  hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
  bad = 0; for (float x : h) bad += (x != 53.f * SPLITS);
  printf("  atomic_mfma check: %g wrong of %zu\n", bad, h.size());
  return 0;
}
