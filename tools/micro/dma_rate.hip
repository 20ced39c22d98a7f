// Micro-benchmark: how fast can one CU pull L2-resident rows into LDS?
//   mode 0: LDS-DMA (buffer_load ... lds, 16 B/lane), row segments of SEG bytes at a row stride of LD bytes
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, same addresses
// Every workgroup (512 threads, one per CU) sweeps its own 1.5 MB panel (L2 resident after the first pass) ITER times.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip ; run: ./dma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int SEG, int MODE, int PERM = 0>
__global__ __launch_bounds__(512) void k(const char* base, int64_t panel_bytes, int ld, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 4 x 32 KiB ring
  const int tid = threadIdx.x, wave = tid >> 6;
  const char* panel = base + (int64_t)blockIdx.x * panel_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)panel, 0, (uint32_t)panel_bytes, 0x00020000);
  constexpr int CPR = SEG / 16;                 // 16-byte chunks per row segment
  const int rows_per_stage = 32768 / SEG;       // one stage = 32 KiB
  const int nstage = (int)(panel_bytes / ((int64_t)rows_per_stage * ld));
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < nstage; ++s) {
      char* dst = lds + (s & 3) * 32768;
#pragma unroll
      for (int i = 0; i < 4; ++i) {            // 2048 chunks per stage / 512 threads
        const int p = i * 512 + tid;
        const int row = p / CPR;
        int c = p % CPR;
        if (PERM == 1) c ^= (row >> 1) & (CPR - 1);          // the lanes of a segment fetch its 16-byte chunks in XOR-permuted order (swizzled LDS image)
        if (PERM == 2) c ^= ((row >> 1) & (CPR - 1)) & ~1;   // permuted in aligned 32-byte pairs only
        if (PERM == 3) c ^= ((row >> 1) & (CPR - 1)) & ~3;   // permuted in aligned 64-byte quads only
        const uint32_t voff = (uint32_t)((s * rows_per_stage + row) * ld + c * 16);
        if (MODE == 0) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + (i * 512 + wave * 64) * 16), 16, voff, 0, 0, 0);
        } else {
          const i32x4 v = *(const i32x4*)(panel + voff);
          *(i32x4*)(dst + p * 16) = v;
        }
      }
      if (MODE == 0 && (s & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // keep ~2 stages in flight
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc += ((int*)lds)[tid];
  if (acc == 0x7fffffff) sink[0] = acc;
}

template <int SEG, int MODE, int PERM = 0>
static void run(const char* name, char* buf, int64_t panel, int ld, int* sink) {
  const int iters = 40, grid = 256;
  hipFuncSetAttribute((const void*)k<SEG, MODE, PERM>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  k<SEG, MODE, PERM><<<grid, 512, 131072>>>(buf, panel, ld, 2, sink);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<SEG, MODE, PERM><<<grid, 512, 131072>>>(buf, panel, ld, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int rows_per_stage = 32768 / SEG;
  const int nstage = (int)(panel / ((int64_t)rows_per_stage * ld));
  const double bytes = (double)grid * iters * nstage * 32768.0;
  printf("%-44s %7.2f TB/s  %6.1f B/clk/CU (at 2.4 GHz)\n", name, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  const int64_t panel = 3 << 19;   // 1.5 MiB per workgroup -> 384 MiB total? no: 256 x 1.5 MiB = 384 MiB (beyond L2, inside the Infinity Cache + HBM)
  char* buf; int* sink;
  hipMalloc(&buf, 256 * panel); hipMalloc(&sink, 4);
  hipMemset(buf, 1, 256 * panel);
  run<64, 0>("LDS-DMA, 64-B segments, stride 1536 B", buf, panel, 1536, sink);
  run<128, 0>("LDS-DMA, 128-B segments, stride 1536 B", buf, panel, 1536, sink);
  run<256, 0>("LDS-DMA, 256-B segments, stride 1536 B", buf, panel, 1536, sink);
  run<1024, 0>("LDS-DMA, contiguous (1 KiB rows, stride 1 KiB)", buf, panel, 1024, sink);
  run<64, 0, 1>("LDS-DMA, 64-B segments, chunks XOR-permuted", buf, panel, 1536, sink);
  run<128, 0, 1>("LDS-DMA, 128-B segments, chunks XOR-permuted", buf, panel, 1536, sink);
  run<128, 0, 2>("LDS-DMA, 128-B segments, permuted in 32-B pairs", buf, panel, 1536, sink);
  run<128, 0, 3>("LDS-DMA, 128-B segments, permuted in 64-B quads", buf, panel, 1536, sink);
  run<64, 1>("VGPR path, 64-B segments, stride 1536 B", buf, panel, 1536, sink);
  run<128, 1>("VGPR path, 128-B segments, stride 1536 B", buf, panel, 1536, sink);
  run<1024, 1>("VGPR path, contiguous", buf, panel, 1024, sink);
  // small panels: everything L2 resident (64 KiB per workgroup)
  const int64_t small = 1 << 16;
  run<64, 0>("L2-resident LDS-DMA, 64-B segments", buf, small, 1536 / 24, sink);
  return 0;
}
