// K2/K4/K6 second generation: bf16 MFMA GEMM with a multi-stage LDS ring kept in flight ACROSS barriers.
//
// v1 (gemm_bf16.hip) has one 64-deep K tile in flight per workgroup and drains it (vmcnt(0)) at every
// barrier.  Here:
//   * BK = 32 stages in a ring of STAGES LDS buffers; STAGES-1 stages are always in flight.  The wait
//     that retires stage t is a COUNTED `s_waitcnt vmcnt(n)` followed by a raw `s_barrier` (never
//     __syncthreads(), which would drain the LDS-DMA queue), then stage t+STAGES-1 is issued into the
//     buffer every wave finished reading one iteration ago.
//   * Variants (make_plan picks per shape; what is actually used is listed in DESIGN.md section 4):
//       256x256, 8 waves, 1 workgroup/CU  "ping-pong": the two waves of every SIMD alternate a LOAD slot
//                (fragment reads + DMA issue + counted wait) and an MFMA slot (32 MFMAs), one slot out of
//                phase.  QKV / FFN-up forward, FFN-down dgrad, and - with both operands read transposed -
//                the QKV / FFN weight gradients (split-K).
//       256x128 / 128x128, 4 waves, 2 workgroups/CU  plain ring / ring with double-buffered fragments
//                (kept for the fp32-master fallback layouts and for experiments; v1 wins on N = 768).
//       gemm2p_kernel  persistent ping-pong: experiment builds only (`make diag`, -DNBEST_EXPERIMENTS), see there.
//   * LDS images (LDS-DMA is lane-linear, so the swizzle lives in the SOURCE address and the read
//     address): k-contiguous operand [rows][32 k], 64-B rows: chunk ^= (-(row>>2))&3 (conflict-free
//     for the 16-lane groups of ds_read_b128 even though a group mixes two k-chunks);
//     k-strided operand [32 k][cols]: chunk ^= 2*(k&3) + 8*((k>>3)&1) for ds_read_b64_tr_b16.
//   * Transposed reads go through inline asm in the ping-pong loop: the builtin makes the compiler drain
//     the whole DMA ring (`s_waitcnt vmcnt(0)`) in front of every such read (see ds_read_tr_asm).
//   * epilogue as v1 (fp32 restage through wave-private LDS into a row-contiguous layout, 16-byte
//     whole-line I/O, fast erf) but in 32-row chunks, with the residual / pre-activation rows of the
//     next chunk prefetched while the current one is processed.
//   * -DNBEST_DIAG=<mask> (`make diag DIAG=<mask>`) builds are timing-only ablations / cycle-stamp builds (tools/,
//     profiles/README.md); the shipped library is built with NBEST_DIAG = 0 and without NBEST_EXPERIMENTS.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int BK = 32;
#ifndef NBEST_DIAG
#define NBEST_DIAG 0
#endif
constexpr int DIAG = NBEST_DIAG;   // timing-only ablation builds of the ping-pong loop: 1 no in-loop DMA, 2 no fragment reads, 4 no MFMA

struct GemmP2 {
  const bf16* A; const bf16* B; void* C; const float* bias; const bf16* R; bf16* U; float* slab; float* colpart;
  int64_t M, N, K, lda, ldb, ldc, ldr, ldu;
  int64_t k_per_split;
  int tiles_m, tiles_n, splits, accumulate;
  uint32_t a_bytes, b_bytes;
  uint32_t c_bytes, r_bytes, u_bytes;   // extents of the M valid rows of C / R / U (range-checked buffer access of the register epilogue)
  DropCfg drop;
  int stream_out;     // epilogue stores are nontemporal (common.h st_stream)
  int gn;             // tile columns per L2 group (common.h nb_tile_coords)
  // second problem of a weight-gradient PAIR (nbest_wgrad_pair): output rows >= m_split of the virtual [M][N] result are
  // A2^T . B2 (same K and N); m_split = 0: none
  const bf16* A2; const bf16* B2;
  int64_t lda2, ldb2, m_split;
  uint32_t a2_bytes, b2_bytes;
  // B pre-packed for this kernel's tile (nbest_pack_weights): the LDS image of every (tile column, K stage) stored contiguously;
  // nullptr: B is read row by row
  const bf16* Bp;
  uint32_t bp_bytes;
  int bp_bn;          // tile width B was packed for (nbest_pack_bn): the kernel's BN, or BN / 2 (a 384-column tile reads two 192-column blocks)
};

__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// one operand tile: ROWS x 32 bf16 = ROWS*4 16-byte chunks, 256 threads -> ROWS/64 DMA instructions per thread
// PW > 0 (k-contiguous B operand of the register-epilogue kernels): within every block of PW rows, LDS row 16 j + c is fetched
// from operand row (PW / 16) * c + j.
// ROWS * 4 chunks over NT threads need not divide (192-row B tile, 512 threads: 1.5 instructions per thread): the last instruction
// is issued by EVERY wave so that all waves count the same number of LDS-DMA operations per stage (the counted s_waitcnt vmcnt of
// the ring assume it); the waves with no chunk left read past the end of the buffer (range check: zero fill, no memory traffic)
// into their own 1-KiB slot of `dump`.
template <bool TR, int ROWS, int NT, int AUX = 0, int PW = 0>
__device__ __forceinline__ void stage_tile2(__amdgpu_buffer_rsrc_t rs, char* tile, int64_t row0, int64_t k0, int64_t ld, int tid,
                                            char* dump = nullptr) {
  const int wave = tid >> 6;
  constexpr int NI = (ROWS * 4 + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = i * NT + tid;
    uint32_t voff;
    if (!TR) {
      const int row = p >> 2, slot = p & 3;
      const int kc = slot ^ ((-(row >> 2)) & 3);
      int grow = row;
      if constexpr (PW > 0) { const int x = row % PW; grow = row - x + (PW / 16) * (x & 15) + (x >> 4); }
      voff = (uint32_t)(((row0 + grow) * ld + k0 + kc * 8) * 2);

    } else {
      constexpr int CPR = ROWS / 8;  // 16-byte chunks per k-row
      const int krow = p / CPR, slot = p % CPR;
      const int mc = slot ^ (2 * (krow & 3) + 8 * ((krow >> 3) & 1));
      voff = (uint32_t)(((k0 + krow) * ld + row0 + mc * 8) * 2);
    }
    char* dst = tile + (i * NT + wave * 64) * 16;
    if constexpr ((ROWS * 4) % NT != 0) {
      if (i == NI - 1 && (i * NT + wave * 64) * 16 >= ROWS * 64) {   // wave-uniform: this wave has no chunk in the last instruction
        voff = 0xFFFFFFF0u;
        dst = dump + (wave % (NT / 64)) * 1024;
      }
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, voff, 0, 0, AUX);
  }
}

// the same tile from a PRE-PACKED operand: the LDS image of every (tile, stage) stored contiguously, so the LDS-DMA is a linear copy
// (1 KiB contiguous per wave-instruction instead of sixteen 64-byte row segments)
template <int ROWS, int NT, int AUX = 0>
__device__ __forceinline__ void stage_tile_packed(__amdgpu_buffer_rsrc_t rs, char* tile, uint32_t stage_byte0, int tid, char* dump = nullptr) {
  const int wave = tid >> 6;
  constexpr int NI = (ROWS * 4 + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    uint32_t voff = stage_byte0 + (uint32_t)(i * NT + tid) * 16u;
    char* dst = tile + (i * NT + wave * 64) * 16;
    if constexpr ((ROWS * 4) % NT != 0) {
      if (i == NI - 1 && (i * NT + wave * 64) * 16 >= ROWS * 64) {
        voff = 0xFFFFFFF0u;
        dst = dump + (wave % (NT / 64)) * 1024;
      }
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, voff, 0, 0, AUX);
  }
}

// A operand of the k-contiguous ping-pong kernels, staged in PAIRS of K stages: one LDS-DMA wave-instruction fetches 8 rows x 128
// contiguous bytes (two stages' worth of a row) instead of 16 rows x 64 - LDS-DMA delivers 33 B/clk/CU on 128-byte segments against 22
// on 64-byte ones (tools/micro/dma_rate.hip), and the operand delivery is co-critical with the MFMA issue in these kernels.  Pair image:
// [ROWS][128 B], 16-byte chunk c = 4 * (stage & 1) + kc of a row stored at chunk slot c ^ ((row >> 1) & 7): conflict-free for the
// fragment reads of either stage (a 16-lane group of ds_read_b128 holds 16 distinct rows, 8 with kc and 8 with kc ^ 1).
// I0 .. I1: the wave-instructions issued by this call (a pair is issued in two halves, one per slot of the ping-pong loop)
template <int ROWS, int NT, int AUX, int I0, int I1>
__device__ __forceinline__ void stage_a_pair(__amdgpu_buffer_rsrc_t rs, char* pair, int64_t row0, int64_t k0, int64_t ld, int tid) {
  const int wave = tid >> 6;
  static_assert((ROWS * 8) % NT == 0 && I1 <= (ROWS * 8) / NT, "stage_a_pair: whole instructions only");
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    const int p = i * NT + tid;
    const int row = p >> 3, slot = p & 7;
    const int c = slot ^ ((row >> 1) & 7);
    const uint32_t voff = (uint32_t)(((row0 + row) * ld + k0 + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(pair + (i * NT + wave * 64) * 16), 16, voff, 0, 0, AUX);
  }
}
__device__ __forceinline__ bf16x8 read_frag_pair(const char* pair, int half, int row_base, int lane) {
  const int row = row_base + (lane & 15);
  const int c = 4 * half + (lane >> 4);
  return *(const bf16x8*)(pair + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}
// s_waitcnt vmcnt(n) for a wave-uniform even n <= 12
__device__ __forceinline__ void wait_vm_even(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

// ds_read_b64_tr_b16 through inline asm.  The builtin is treated as a read of unknown LDS memory: with LDS-DMA
// loads still in flight (the whole point of the ring) the compiler puts `s_waitcnt vmcnt(0)` in front of it and
// drains the ring at every stage.  As opaque asm it is left alone; the caller waits with lgkm_wait_tied() before
// the first use of the fragments.
__device__ __forceinline__ bf16x4 ds_read_tr_asm(const char* addr) {
  bf16x4 v;
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)addr;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}

// 16 rows x 32 k fragment of v_mfma_f32_16x16x32_bf16: lane l holds row (l&15), k = 8*(l>>4) + j
template <bool TR, int ROWS, bool ASM = false>
__device__ __forceinline__ bf16x8 read_frag2(const char* tile, int row_base, int lane) {
  if (!TR) {
    const int row = row_base + (lane & 15);
    const int kc = lane >> 4;
    return *(const bf16x8*)(tile + row * 64 + ((kc ^ ((-(row >> 2)) & 3)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    const int col = row_base + 4 * pq;
    bf16x8 out;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int krow = 8 * g + 4 * half + q;
      const int f = 2 * (krow & 3) + 8 * ((krow >> 3) & 1);
      const char* addr = tile + krow * (ROWS * 2) + (((col >> 3) ^ f) << 4) + ((col & 4) ? 8 : 0);
      const bf16x4 v = ASM ? ds_read_tr_asm(addr)
                           : __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)addr);
      out[4 * half + 0] = v[0]; out[4 * half + 1] = v[1]; out[4 * half + 2] = v[2]; out[4 * half + 3] = v[3];
    }
    return out;
  }
}

// Register epilogue of the k-contiguous ("NT") kernels with a bf16 output (kDirect).  The MFMA is issued UN-swapped there
// (D = A . B: a lane holds 4 consecutive M rows 4g + e of ONE output column c = lane & 15 per 16x16 tile), and the B tile is
// staged with its rows PERMUTED: LDS row 16 j + c of a wave's WTN-row block holds the weight row TNt * c + j (the LDS-DMA source
// address is per lane, so the permutation is free and the LDS image, its swizzle and the fragment reads are untouched).  Tile j
// of lane c is then output column TNt * c + j: for a fixed (i, e) a lane owns TNt CONSECUTIVE columns of row 16 i + 4 g + e, and
// one wave-instruction stores 4 rows x (16 lanes x TNt x 2 B) - with 128-column wave tiles (TNt = 8) 16 bytes per lane, 256
// contiguous bytes per row, whole 128-byte lines.  No LDS restage, no barrier, no branch (ragged row tiles: buffer range check).
// (First attempt, kept the MFMA swapped with lane = row: every lane of a store wrote a different row - 64 partial 16-byte
// requests per instruction - and the GEMMs got 5-10 % slower than with the LDS restage.)
template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// helpers of the stripped ping-pong loop (gemm2_kernel, NT == 512 && kDirect): LDS-DMA of one K stage into ring buffer NB_ with
// per-lane source offsets computed once (voA / voB) and the K advance in the scalar offset; fragment reads from ring buffer B_ (a
// template constant: base register + immediate); the MFMA slot between its two workgroup barriers.
template <int NB_, int STAGE, int A_BYTES, int NIA, int NIB, int NT, int BN>
__device__ __forceinline__ void pp_issue(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, char* lds, const uint32_t* voA,
                                         const uint32_t* voB, int wv, bool dumpB, char* dump, uint32_t sA, uint32_t sB) {
#pragma unroll
  for (int i = 0; i < NIA; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(lds + NB_ * STAGE + (i * NT + wv * 64) * 16), 16, voA[i], sA, 0, NB_AUX_A);
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    char* dst = lds + NB_ * STAGE + A_BYTES + (i * NT + wv * 64) * 16;
    const bool d = ((BN * 4) % NT != 0) && i == NIB - 1 && dumpB;      // wave-uniform
    if (d) dst = dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(dst), 16, voB[i], d ? 0u : sB, 0, NB_AUX_B);
  }
}
template <int B_, int STAGE, int A_BYTES, int BM, int BN, int TMt, int TNt>
__device__ __forceinline__ void pp_frags(const char* lds, bf16x8* af, bf16x8* bfr, int row0, int col0, int lane) {
  const char* cur = lds + B_ * STAGE;
#pragma unroll
  for (int j = 0; j < TNt; ++j) bfr[j] = read_frag2<false, BN>(cur + A_BYTES, col0 + j * 16, lane);
#pragma unroll
  for (int i = 0; i < TMt; ++i) af[i] = read_frag2<false, BM>(cur, row0 + i * 16, lane);
}
template <int TMt, int TNt>
__device__ __forceinline__ void pp_mfma_slot(const bf16x8* af, const bf16x8* bfr, f32x4 (&acc)[TMt][TNt]) {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int i = 0; i < TMt; ++i)
#pragma unroll
    for (int j = 0; j < TNt; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// ---- helpers of the stripped ping-pong loop of the weight-gradient (TN) kernel: both operands token-major, fragments by transposed reads.
// LDS image of one operand ring (STAGES stages of ROWS x 32 k): [16 k-row pairs][STAGES][2 k-rows x ROWS * 2 B] - the ring index sits
// BELOW the k-row pair, so that (a) an LDS-DMA wave-instruction (1 KiB = 2 k-rows of 256 features) still writes contiguously and (b) the
// stage offset (1 KiB x stage) and the upper half's offset of a fragment fit the 16-bit immediate of ds_read: a fragment address is ONE
// per-lane register computed before the loop + an immediate.  Inside a k-row the 16-byte chunk swizzle of stage_tile2<true> is unchanged
// (a k-row starts at a multiple of 512 B in either layout: same banks).
template <int OFF> __device__ __forceinline__ bf16x4 ds_read_tr_off(uint32_t a) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
template <int B_, int STAGES, int NF>
__device__ __forceinline__ void tt_frags(const uint32_t* base, bf16x8* fr) {
  constexpr int HALF = 2 * STAGES * 1024;      // k-rows + 4: two k-row pairs further
#pragma unroll
  for (int t = 0; t < NF; ++t) {
    const bf16x4 lo = ds_read_tr_off<B_ * 1024>(base[t]);
    const bf16x4 hi = ds_read_tr_off<B_ * 1024 + HALF>(base[t]);
    fr[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}
// LDS-DMA of one stage of both operands into ring slot NB_: wave wv issues pieces (i * 8 + wv), i = 0, 1, of each operand
template <int NB_, int STAGES>
__device__ __forceinline__ void tt_issue(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, char* ldsA, char* ldsB, const uint32_t* voA,
                                         const uint32_t* voB, int wv) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(ldsA + ((i * 8 + wv) * STAGES + NB_) * 1024), 16, voA[i], 0, 0, NB_AUX_A);
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(ldsB + ((i * 8 + wv) * STAGES + NB_) * 1024), 16, voB[i], 0, 0, NB_AUX_B);
}
// end of a LOAD slot: the (asm) fragment reads have landed; every fragment is redefined AFTER the wait so that no MFMA can be scheduled above it
template <int TMt, int TNt>
__device__ __forceinline__ void tt_frags_ready(bf16x8* af, bf16x8* bfr) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < TMt; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
  for (int j = 0; j < TNt; ++j) asm volatile("" : "+v"(bfr[j]));
}
template <int TMt, int TNt>
__device__ __forceinline__ void tt_mfma(const bf16x8* af, const bf16x8* bfr, f32x4 (&acc)[TMt][TNt]) {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int i = 0; i < TMt; ++i)
#pragma unroll
    for (int j = 0; j < TNt; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
}

#define NB_STAMP(IDX)                                                                                          \
  if ((DIAG & 32) && p.U && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 700))                       \
    ((uint64_t*)p.U)[16000 + (blockIdx.x ? 8 : 0) + (IDX)] = __builtin_readcyclecounter()

template <int BM, int BN, int WM, int WN, int STAGES, bool TA, bool TB, int EPI, bool SYM = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm2_kernel(GemmP2 p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  NB_STAMP(0);
  constexpr int NT = WM * WN * 64;                   // 256 threads (2 x 2 waves) or 512 (2 x 4 waves)
  constexpr bool PIPE = (BM == 128 && NT == 256);   // second fragment register set: fits only the 64x64 wave tile
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TMt = WTM / 16, TNt = WTN / 16;
  // k-contiguous operands with a bf16 output: un-swapped MFMA, permuted B rows, epilogue straight from the accumulators (see above)
  constexpr bool kDirect = (!TA && !TB && EPI != NBEST_EPI_F32_SPLITK);
  constexpr int kPW = kDirect ? WTN : 0;
  static_assert(kDirect ? (WTN == 64 || WTN == 96 || WTN == 128) : WTN == 64, "LDS-restaged epilogue assumes 64-column wave tiles");
  static_assert(TNt != 6 || (EPI != NBEST_EPI_BIAS_GELU && EPI != NBEST_EPI_DGELU), "96-column wave tiles: no GELU epilogues (8-bit rows)");
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NDMA = (BM * 4 + NT - 1) / NT + (BN * 4 + NT - 1) / NT;   // LDS-DMA instructions per thread and stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int nwg = gridDim.x;
  const int id = xcd_remap2(blockIdx.x, nwg);
  const int tiles = p.tiles_m * p.tiles_n;
  const int z = id / tiles, t = id - z * tiles;
  int tile_m, tile_n;
  nb_tile_coords(t, p.tiles_m, p.gn, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
  const int64_t kbeg = (int64_t)z * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nk = (int)((kend - kbeg + BK - 1) / BK);

  // operands of this tile: the second problem's for the tiles below m_split of a weight-gradient pair (workgroup-uniform)
  const bf16* opA = p.A; const bf16* opB = p.B;
  int64_t lda_ = p.lda, ldb_ = p.ldb, m0a = m0;
  uint32_t a_bytes = p.a_bytes, b_bytes = p.b_bytes;
  if constexpr (TA && TB && EPI == NBEST_EPI_F32_SPLITK) {
    if (p.m_split > 0 && m0 >= p.m_split) {
      opA = p.A2; opB = p.B2; lda_ = p.lda2; ldb_ = p.ldb2; m0a = m0 - p.m_split; a_bytes = p.a2_bytes; b_bytes = p.b2_bytes;
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)opA, 0, a_bytes, 0x00020000);
  const bool b_packed = kDirect && NT == 512 && p.Bp != nullptr && ((BN != 384 && BN != 512) || !(DIAG & 0x8000));      // workgroup-uniform
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(b_packed ? (void*)p.Bp : (void*)opB, 0, b_packed ? p.bp_bytes : b_bytes, 0x00020000);
  const int nkt = (int)(p.K / BK);

  // epilogue operands that do not depend on the K loop
  constexpr bool kHasBias = (EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU || EPI == NBEST_EPI_BIAS_DROP_RES);
  constexpr bool kHasR = (EPI == NBEST_EPI_BIAS_DROP_RES || EPI == NBEST_EPI_RES);
  constexpr bool kHasUin = (EPI == NBEST_EPI_DGELU);
  constexpr bool kPre = kHasR || kHasUin;
  const int64_t en8 = n0 + wn * WTN + (lane & 7) * 8;
  const int64_t erow0 = m0 + wm * WTM + (lane >> 3);
  // residual rows are bf16 (16 bytes per lane); GELU' rows are 8-bit (gd_pack4: 8 bytes per lane, in .x/.y)
  auto load_pre = [&](int64_t m) -> i32x4 {
    if constexpr (kHasR) {
      return (m < p.M) ? *(const i32x4*)(p.R + m * p.ldr + en8) : i32x4{0, 0, 0, 0};
    } else {
      const i32x2 q = (m < p.M) ? *(const i32x2*)((const uint8_t*)p.U + m * p.ldu + en8) : i32x2{0, 0};
      return i32x4{q[0], q[1], 0, 0};
    }
  };
  f32x4 pb0 = {0, 0, 0, 0}, pb1 = {0, 0, 0, 0};
  if (kHasBias && !kDirect) { pb0 = *(const f32x4*)(p.bias + en8); pb1 = *(const f32x4*)(p.bias + en8 + 4); }
  i32x4 pre[TMt / 2][4];   // chunk 0 is fetched before the K loop, the others right after it (fragment registers are dead then)
  if (kPre && !kDirect) {
#pragma unroll
    for (int it = 0; it < 4; ++it) pre[0][it] = load_pre(erow0 + it * 8);
  }
  // register epilogue (kDirect): lane (dg, dc) owns, for every (i, e), the TNt consecutive columns dcol .. dcol + TNt - 1 of row
  // drow0 + 16 i + e.  Output, residual and GELU' rows go through range-checked buffer descriptors (rows >= M: stores dropped,
  // loads return zero): no branch anywhere.
  const int dg = lane >> 4, dc = lane & 15;
  const int64_t dcol = n0 + wn * WTN + TNt * dc;
  const int64_t drow0 = m0 + wm * WTM + 4 * dg;
  constexpr int NPRE0 = SYM ? 0 : 1;   // 16-row tiles whose residual / GELU' rows are fetched before the K loop (the rest right after it)
  float db[TNt];
  u32x4 dpre[TMt][4];        // TNt = 8: 16 bytes of residual (8 bf16) per (i, e); TNt = 6: 12 bytes; TNt = 4: 8 bytes; GELU': TNt bytes
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.R, 0, p.r_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, p.u_bytes, 0x00020000);
  const uint32_t voR = (uint32_t)((drow0 * p.ldr + dcol) * 2), voU = (uint32_t)(drow0 * p.ldu + dcol);
  const uint32_t svR = (uint32_t)(2 * p.ldr), svU = (uint32_t)p.ldu;   // bytes per row
  auto load_dpre = [&](int i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t rr = (uint32_t)(16 * i + e);
      if constexpr (kHasR) {
        if constexpr (TNt == 8) dpre[i][e] = __builtin_amdgcn_raw_buffer_load_b128(rsR, voR + rr * svR, 0, 0);
        else if constexpr (TNt == 6) { const u32x3 q = __builtin_amdgcn_raw_buffer_load_b96(rsR, voR + rr * svR, 0, 0); dpre[i][e] = u32x4{q[0], q[1], q[2], 0}; }
        else { const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rsR, voR + rr * svR, 0, 0); dpre[i][e] = u32x4{q[0], q[1], 0, 0}; }
      } else if constexpr (kHasUin) {
        if constexpr (TNt == 8) { const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rsU, voU + rr * svU, 0, 0); dpre[i][e] = u32x4{q[0], q[1], 0, 0}; }
        else dpre[i][e] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rsU, voU + rr * svU, 0, 0), 0, 0, 0};
      }
    }
  };
  auto load_bias = [&]() {
    if (kHasBias) {
#pragma unroll
      for (int q = 0; q < TNt / 2; ++q) {
        const f32x2 b2 = *(const f32x2*)(p.bias + dcol + 2 * q);
        db[2 * q] = b2[0]; db[2 * q + 1] = b2[1];
      }
    }
  };
  if constexpr (kDirect) {
    if constexpr (!SYM) load_bias();
    if (kHasR || kHasUin) {
#pragma unroll
      for (int i = 0; i < NPRE0; ++i) load_dpre(i);
    }
  }

  f32x4 acc[TMt][TNt];
#pragma unroll
  for (int i = 0; i < TMt; ++i)
#pragma unroll
    for (int j = 0; j < TNt; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  if constexpr (SYM) {
    // ---- symmetric software pipeline (experiment): every wave interleaves the fragment reads of stage kt+1 (second register set)
    // with the MFMAs of stage kt; ONE workgroup barrier per stage (stage kt+1 landed for everyone / slot of stage kt-1 free).
    static_assert(NT == 512 && kDirect && !TA && !TB, "SYM: k-contiguous 8-wave tiles only");
    // The current stage lives in registers only: after the barrier of stage kt every wave has its fragments of stage kt (lgkmcnt(0)
    // before the barrier), so slot kt % STAGES is refilled with stage kt + STAGES right away - three stages stay in flight.
#pragma unroll
    for (int s0 = 0; s0 < STAGES; ++s0) {
      if (s0 < nk) {
        stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + s0 * STAGE, m0a, kbeg + (int64_t)s0 * BK, lda_, tid);
        stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + s0 * STAGE + A_BYTES, n0, kbeg + (int64_t)s0 * BK, ldb_, tid, lds + STAGES * STAGE);
      }
    }
    {
      const int younger = (nk - 1 < STAGES - 1) ? nk - 1 : STAGES - 1;
      if (younger >= 3) wait_vm<3 * NDMA>();
      else if (younger == 2) wait_vm<2 * NDMA>();
      else if (younger == 1) wait_vm<NDMA>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 aP[TMt], bP[TNt], aQ[TMt], bQ[TNt];
#pragma unroll
    for (int j = 0; j < TNt; ++j) bP[j] = read_frag2<false, BN>(lds + A_BYTES, wn * WTN + j * 16, lane);
#pragma unroll
    for (int i = 0; i < TMt; ++i) aP[i] = read_frag2<false, BM>(lds, wm * WTM + i * 16, lane);
    int buf = 0;
    // body of a stage that has a successor: wait for / publish stage kt+1, refill the ring, then ONE basic block of 32 MFMAs on the
    // current fragments with the 12 reads of the next fragments spread between them
    auto body = [&](int kt, bf16x8 (&aX)[TMt], bf16x8 (&bX)[TNt], bf16x8 (&aY)[TMt], bf16x8 (&bY)[TNt]) {
      // issued so far: stages .. kt+3 (those below nk); stage kt+1 must have landed, this wave's fragments of stage kt too
      if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NDMA) : "memory");
      else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + STAGES < nk && !(DIAG & 256)) {
        const int64_t k0 = kbeg + (int64_t)(kt + STAGES) * BK;
        stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + buf * STAGE, m0a, k0, lda_, tid);
        stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + buf * STAGE + A_BYTES, n0, k0, ldb_, tid, lds + STAGES * STAGE);
      }
      int nx = buf + 1;
      if (nx >= STAGES) nx -= STAGES;
      const char* nxt = lds + nx * STAGE;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(DIAG & 128)) {
#pragma unroll
        for (int j = 0; j < TNt; ++j) bY[j] = read_frag2<false, BN>(nxt + A_BYTES, wn * WTN + j * 16, lane);
#pragma unroll
        for (int i = 0; i < TMt; ++i) aY[i] = read_frag2<false, BM>(nxt, wm * WTM + i * 16, lane);
      } else {
#pragma unroll
        for (int j = 0; j < TNt; ++j) bY[j] = bX[j];
#pragma unroll
        for (int i = 0; i < TMt; ++i) aY[i] = aX[i];
      }
      if constexpr (!(DIAG & 64)) {
#pragma unroll
      for (int i = 0; i < TMt; ++i)
#pragma unroll
        for (int j = 0; j < TNt; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aX[i], bX[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < TMt; ++i) asm volatile("" :: "v"(aX[i]));
#pragma unroll
        for (int j = 0; j < TNt; ++j) asm volatile("" :: "v"(bX[j]));
      }
      if constexpr (!(DIAG & (64 | 128 | 512))) {
      constexpr int N3 = TMt * TNt - 2 * (TMt + TNt), N2 = TMt + TNt - N3;
      static_assert(N3 >= 0 && N2 >= 0, "SYM: interleave pattern");
#pragma unroll
      for (int g = 0; g < N3; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int g = 0; g < N2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      buf = nx;
    };
    // (the body of the LAST stage runs the same code: its barrier is harmless and its fragment reads fetch a stale slot nobody uses -
    // a separate tail would be a third copy of the body, and the register allocator spilled 300 dwords per thread around it)
    for (int kt = 0; kt < nk; kt += 2) {
      body(kt, aP, bP, aQ, bQ);
      if (kt + 1 < nk) body(kt + 1, aQ, bQ, aP, bP);
    }
    load_bias();
  } else if constexpr (NT == 512 && kDirect && BM == 256 && (DIAG & 2048)) {
    // ---- EXPERIMENT (`make diag DIAG=2048`; measured 25 % SLOWER than per-stage staging - waves wait twice as long for the LDS-DMA,
    // no LDS bank conflicts; cause not found - kept for the next attempt) ----
    // ---- ping-pong as below, with the A operand staged in PAIRS of stages (128-byte row segments, stage_a_pair): LDS = 3 pair buffers
    // of A (2 x A_BYTES each) | STAGES tiles of B | dump slots.  A LOAD slot is as long as an MFMA slot and most of it is LDS-DMA issue,
    // so a pair is issued in two halves, one per slot: slot j issues half (j & 1) of A pair j/2 + 2 and then B stage j + 3 - four
    // wave-instructions per slot, as with per-stage staging.  Per wave the issue order is  A0 A1 B0 B1 B2 | A2a B3 | A2b B4 | A3a B5 | ...
    // and slot kt retires this wave's share of stage kt+1 = B(kt+1), the last instruction of slot kt-2 (A((kt+1)/2) is older): whatever
    // slots kt-1 and kt issued may stay in flight.
    constexpr int AP_BYTES = 2 * A_BYTES, NAP = 3;
    constexpr int NDA = (BM * 8) / NT, NDB = (BN * 4 + NT - 1) / NT, NDH = NDA / 2;
    static_assert(NDH % 2 == 0 && NDB % 2 == 0 && 2 * NDH + 2 * NDB <= 12, "wait_vm_even covers even counts up to 12");
    static_assert(STAGES == 4, "pair path: B ring of 4");
    char* const ldsB = lds + NAP * AP_BYTES;
    char* const dumpB = ldsB + STAGES * B_BYTES;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    auto issue_b = [&](int st_) {
      char* dst = ldsB + (st_ & (STAGES - 1)) * B_BYTES;
      if (b_packed) stage_tile_packed<BN, NT, NB_AUX_B>(rsB, dst, (uint32_t)((tile_n * nkt + st_) * B_BYTES), tid, dumpB);
      else stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, dst, n0, kbeg + (int64_t)st_ * BK, ldb_, tid, dumpB);
    };
    // slot j: half (j & 1) of the pair holding stages j' + 4, j' + 5 (j' = j rounded down to even); exists iff stage j' + 4 does
    auto a_half_exists = [&](int j) { return (j & ~1) + 4 < nk; };
    auto issue_a_half = [&](int j) {
      const int q = (j >> 1) + 2;
      char* dst = lds + (q % NAP) * AP_BYTES;
      const int64_t k0 = kbeg + (int64_t)q * (2 * BK);
      if (j & 1) stage_a_pair<BM, NT, NB_AUX_A, NDH, NDA>(rsA, dst, m0a, k0, lda_, tid);
      else stage_a_pair<BM, NT, NB_AUX_A, 0, NDH>(rsA, dst, m0a, k0, lda_, tid);
    };
    if (0 < nk) stage_a_pair<BM, NT, NB_AUX_A, 0, NDA>(rsA, lds, m0a, kbeg, lda_, tid);
    if (2 < nk) stage_a_pair<BM, NT, NB_AUX_A, 0, NDA>(rsA, lds + AP_BYTES, m0a, kbeg + 2 * BK, lda_, tid);
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
      if (s0 < nk) issue_b(s0);
    {
      const int younger = (nk - 1 < STAGES - 2) ? nk - 1 : STAGES - 2;   // B stages issued after B0
      wait_vm_even(younger * NDB);
    }
    __builtin_amdgcn_s_barrier();                 // stage 0 landed for everyone
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();   // offset group 1 by one slot
    bf16x8 af[TMt], bfr[TNt];
    for (int kt = 0; kt < nk; ++kt) {
      // ---------------- LOAD slot ----------------
      const bool ah = a_half_exists(kt), bh = kt + STAGES - 1 < nk;
      if (ah) issue_a_half(kt);
      if (bh) issue_b(kt + STAGES - 1);
      {
        const char* curB = ldsB + (kt & (STAGES - 1)) * B_BYTES;
        const char* curA = lds + ((kt >> 1) % NAP) * AP_BYTES;
#pragma unroll
        for (int j = 0; j < TNt; ++j) bfr[j] = read_frag2<false, BN>(curB, wn * WTN + j * 16, lane);
#pragma unroll
        for (int i = 0; i < TMt; ++i) af[i] = read_frag_pair(curA, kt & 1, wm * WTM + i * 16, lane);
      }
      if (kt + 1 < nk) {   // retire this wave's share of stage kt+1
        int cnt = (ah ? NDH : 0) + (bh ? NDB : 0);                                                    // this slot
        if (kt == 0) cnt += (2 < nk ? NDB : 0);                                                      // B2 of the prologue follows B1
        else cnt += (a_half_exists(kt - 1) ? NDH : 0) + (kt + 2 < nk ? NDB : 0);                      // slot kt-1
        wait_vm_even(cnt);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA slot ----------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TMt; ++i)
#pragma unroll
        for (int j = 0; j < TNt; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the barrier count
  } else if constexpr (NT == 512 && kDirect && !(DIAG & 0x8000)) {
    // ---- ping-pong as in the generic branch below (two wave groups one slot out of phase, LOAD slot | barrier | MFMA slot | barrier),
    // with the LOAD slot stripped to its memory instructions (round 4).  The ISA of the generic branch spent, per slot: ~20 VALU
    // instructions on addresses (per-lane DMA source offsets re-derived from k0, LDS destinations computed in VGPRs and moved to M0
    // through v_readfirstlane, fragment addresses from the runtime ring index), and half a dozen scalar branches on conditions that
    // are constant through the steady state (is a stage left to issue, how many may stay in flight, packed or plain B).  On a SIMD
    // shared with a wave in its MFMA slot every VALU instruction of the loading wave takes a vector-issue slot from the MFMAs
    // (MI355X_MICROARCH.md "Two waves per SIMD" item 2): the MFMA slot ran at 20 cycles per 16-cycle MFMA.  Here:
    //   * the wave index is an SGPR (readfirstlane once): every LDS-DMA destination is scalar arithmetic -> s_mov m0;
    //   * the per-lane DMA source offsets are computed ONCE (k0 = 0); the K advance travels in the instruction's scalar offset
    //     (soffset; the range check that zero-fills rows >= M works on the per-lane offset, which holds the row);
    //   * the ring index is a template constant inside a body unrolled over the ring: ds_read_b128 addresses are one per-lane base
    //     plus an immediate;
    //   * steady state (a stage to issue in every slot, two stages in flight behind the wait) and tail are separate code.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int grp = wv >> 2;
    constexpr int NIA = (BM * 4) / NT, NIB = (BN * 4 + NT - 1) / NT;
    static_assert((BM * 4) % NT == 0, "A tile: whole LDS-DMA instructions");
    uint32_t voA[NIA], voB[NIB];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int pp = i * NT + tid, row = pp >> 2, slot = pp & 3, kc = slot ^ ((-(row >> 2)) & 3);
      voA[i] = (uint32_t)(((m0a + row) * lda_ + kc * 8) * 2);
    }
    bool dumpB = false;     // this wave has no chunk in the last B instruction of a 192-row tile (wave-uniform)
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int pp = i * NT + tid;
      if (b_packed) {
        // packed image: per (packed tile column, K stage) a block of bp_bn x 32; a tile wider than the packing reads BN / bp_bn blocks of
        // ADJACENT packed tile columns, nkt stage blocks apart
        const int per = p.bp_bn * 4;                          // 16-byte chunks per packed stage block
        const int blk = pp / per;
        voB[i] = (uint32_t)blk * (uint32_t)nkt * (uint32_t)(per * 16) + (uint32_t)(pp - blk * per) * 16u;
      } else {
        const int row = pp >> 2, slot = pp & 3, kc = slot ^ ((-(row >> 2)) & 3);
        int grow = row;
        if constexpr (kPW > 0) { const int x = row % kPW; grow = row - x + (kPW / 16) * (x & 15) + (x >> 4); }
        voB[i] = (uint32_t)(((n0 + grow) * ldb_ + kc * 8) * 2);
      }
      if ((BN * 4) % NT != 0 && i == NIB - 1 && (i * NT + wv * 64) * 16 >= BN * 64) { voB[i] = 0xFFFFFFF0u; dumpB = true; }
    }
    const uint32_t sB0 = b_packed ? (uint32_t)(tile_n * nkt) * (uint32_t)B_BYTES : (uint32_t)(kbeg * 2);
    const uint32_t sBstep = b_packed ? (uint32_t)(p.bp_bn * BK * 2) : (uint32_t)(BK * 2);
    char* const dump = lds + STAGES * STAGE + (wv % (NT / 64)) * 1024;
    // (device function templates, not lambdas: the HOST pass of hipcc silently drops the kernel stub of every instantiation whose body
    // calls a generic lambda that issues LDS-DMA builtins - the library then fails to load with an undefined kernel symbol)
    bf16x8 af[TMt], bfr[TNt];
    const uint32_t sA0 = (uint32_t)(kbeg * 2);
#define PP_ISSUE(ST, NB_) pp_issue<(NB_), STAGE, A_BYTES, NIA, NIB, NT, BN>(rsA, rsB, lds, voA, voB, wv, dumpB, dump, sA0 + (uint32_t)(ST) * (uint32_t)(BK * 2), sB0 + (uint32_t)(ST) * sBstep)
#define PP_FRAGS(B_) pp_frags<(B_), STAGE, A_BYTES, BM, BN, TMt, TNt>(lds, af, bfr, wm * WTM, wn * WTN, lane)
#define PP_MFMA() pp_mfma_slot<TMt, TNt>(af, bfr, acc)
    // steady-state slot kt (ring buffer B_ = kt % STAGES): issue stage kt + STAGES - 1, read the fragments of stage kt, retire this
    // wave's share of stage kt + 1 (STAGES - 2 younger stages stay in flight), then the MFMA slot
#define PP_STEADY(KT, B_)                                  \
    do {                                                   \
      PP_ISSUE((KT) + STAGES - 1, ((B_) + STAGES - 1) % STAGES); \
      PP_FRAGS(B_);                                        \
      wait_vm<(STAGES - 2) * NDMA>();                      \
      PP_MFMA();                                           \
    } while (0)
    // tail / generic slot: runtime conditions as in the generic branch
#define PP_GENERIC(KT, B_)                                                                          \
    do {                                                                                            \
      if ((KT) + STAGES - 1 < nk) PP_ISSUE((KT) + STAGES - 1, ((B_) + STAGES - 1) % STAGES);        \
      PP_FRAGS(B_);                                                                                 \
      const int c_ = (nk - 1 - (KT) < STAGES - 1) ? nk - 1 - (KT) : STAGES - 1;                     \
      if (STAGES >= 5 && c_ >= 4) wait_vm<3 * NDMA>();                                              \
      else if (c_ >= 3) wait_vm<2 * NDMA>();                                                        \
      else if (c_ == 2) wait_vm<NDMA>();                                                            \
      else if (c_ == 1) wait_vm<0>();                                                               \
      PP_MFMA();                                                                                    \
    } while (0)
    // prologue: stages 0 .. STAGES-2 into buffers 0 .. STAGES-2
    if (0 < nk) PP_ISSUE(0, 0);
    if (1 < nk) PP_ISSUE(1, 1 % STAGES);
    if (STAGES > 3 && 2 < nk) PP_ISSUE(2, 2 % STAGES);
    if (STAGES > 4 && 3 < nk) PP_ISSUE(3, 3 % STAGES);
    {
      const int younger = (nk - 1 < STAGES - 2) ? nk - 1 : STAGES - 2;
      if (STAGES >= 5 && younger >= 3) wait_vm<3 * NDMA>();
      else if (younger >= 2) wait_vm<2 * NDMA>();
      else if (younger == 1) wait_vm<NDMA>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();                 // stage 0 landed for everyone
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();   // offset group 1 by one slot
    int kt = 0;
    const int n_steady = nk - (STAGES - 1);        // slots that still have a stage to issue
    for (; kt + STAGES <= n_steady; kt += STAGES) {
      PP_STEADY(kt, 0);
      PP_STEADY(kt + 1, 1);
      PP_STEADY(kt + 2, 2);
      if constexpr (STAGES > 3) PP_STEADY(kt + 3, 3 % STAGES);
      if constexpr (STAGES > 4) PP_STEADY(kt + 4, 4 % STAGES);
    }
    // (kt is a multiple of STAGES here: slot kt + r uses ring buffer r)
    for (; kt < nk; kt += STAGES) {
      PP_GENERIC(kt, 0);
      if (kt + 1 < nk) PP_GENERIC(kt + 1, 1);
      if (kt + 2 < nk) PP_GENERIC(kt + 2, 2);
      if constexpr (STAGES > 3) { if (kt + 3 < nk) PP_GENERIC(kt + 3, 3 % STAGES); }
      if constexpr (STAGES > 4) { if (kt + 4 < nk) PP_GENERIC(kt + 4, 4 % STAGES); }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the barrier count
#undef PP_ISSUE
#undef PP_FRAGS
#undef PP_MFMA
#undef PP_STEADY
#undef PP_GENERIC
  } else if constexpr (NT == 512 && TA && TB && BM == 256 && BN == 256 && STAGES == 4 && !(DIAG & 0x8000)) {
    // ---- the weight-gradient kernel (both operands token-major), round 4.  Two changes against the generic ping-pong branch below:
    // (1) the LOAD slot is stripped to its memory instructions, as in the k-contiguous branch above.  The generic LOAD slot carried 41 vector
    //     and ~19 scalar instructions around its 4 LDS-DMA pieces and 24 transposed reads (per-lane DMA offsets re-derived from k0, LDS
    //     destinations through v_readfirstlane, one XOR + add per fragment address on a runtime ring index); every one of them takes a
    //     vector-issue slot from the wave that shares the SIMD and is in its MFMA slot.  Here: wave index in an SGPR (scalar DMA
    //     destinations), per-lane DMA source offsets advanced by ONE v_add each per slot (the K advance must stay in the range-checked
    //     per-lane offset: the rows past the last token are zero-filled by the descriptor, and a scalar offset is not range-checked),
    //     fragment addresses = 12 per-lane registers set up once + immediates (tt_frags; ring image with the stage index below the
    //     k-row pair), steady state and tail separate.  165 -> 136 us per launch.
    // (2) ONE workgroup barrier per stage instead of two per slot.  Stamps (tools/slot_trace_tt.py) showed LOAD slot 620 and MFMA slot 655
    //     cycles plus 80 - 120 cycles at EACH of the two barriers of a slot - 80 is what s_barrier costs the LAST wave to arrive.  The two
    //     groups need one rendezvous per stage, not four: group 0 runs [LOAD k | MFMA k], group 1 [MFMA k-1 | LOAD k] between two
    //     barriers - each group's LOAD overlaps the other's MFMA by construction, and the barrier at the end of the period is what orders
    //     the ring: every wave has retired its own LDS-DMA share of stage k+1 (counted vmcnt) and its fragment reads of stage k
    //     (lgkmcnt(0) at the END of the LOAD slot) before it, and stage k+1 is read / ring slot k-1 is overwritten only after it.
    //     Measured: NO gain (136 us either way) - kept because it is the simpler contract (one rendezvous, every wait before it).
    // Measured and rejected on this loop (same call, A/B, layer shapes): a fifth ring stage (+0 ... +3 %); two or all four LDS-DMA pieces of a
    // stage issued between the MFMA groups instead of in the LOAD slot (+-0, +2 %); half of the fragment reads issued between the MFMA groups
    // ("rolling" A rows on a five-slot ring, lgkmcnt waits inside the MFMA slot: +10 %).  The stamps leave LOAD 620 + MFMA 655 cycles per wave
    // and stage against 2 x 512 MFMA-pipe cycles per SIMD: 68 % is the ceiling of this two-slot form, 56 % is measured.
    static_assert(WTM == 128 && WTN == 64, "weight-gradient ping-pong: 2 x 4 waves of 128 x 64");
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int grp = wv >> 2;
    char* const ldsA = lds;
    char* const ldsB = lds + STAGES * A_BYTES;
    uint32_t voA[2], voB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pp = i * NT + tid, krow = pp >> 5, slot = pp & 31;
      const int mc = slot ^ (2 * (krow & 3) + 8 * ((krow >> 3) & 1));
      voA[i] = (uint32_t)(((kbeg + krow) * lda_ + m0a + mc * 8) * 2);
      voB[i] = (uint32_t)(((kbeg + krow) * ldb_ + n0 + mc * 8) * 2);
    }
    const uint32_t stepA = (uint32_t)(BK * lda_ * 2), stepB = (uint32_t)(BK * ldb_ * 2);
    uint32_t faA[TMt], faB[TNt];      // fragment addresses (stage 0, lower half)
    {
      const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pq = i16 & 3;
      const int f = 2 * q + 8 * (g & 1);
      const uint32_t rowpart = (uint32_t)((4 * g + (q >> 1)) * (STAGES * 1024) + (q & 1) * 512 + ((pq & 1) ? 8 : 0));
      const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)ldsA;
      const uint32_t b0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)ldsB;
#pragma unroll
      for (int t = 0; t < TMt; ++t) faA[t] = a0 + rowpart + (uint32_t)((((wm * WTM + t * 16) >> 3) + (pq >> 1)) ^ f) * 16u;
#pragma unroll
      for (int t = 0; t < TNt; ++t) faB[t] = b0 + rowpart + (uint32_t)((((wn * WTN + t * 16) >> 3) + (pq >> 1)) ^ f) * 16u;
    }
    bf16x8 af[TMt], bfr[TNt];
    static_assert(NDMA == 4, "weight-gradient ping-pong: two LDS-DMA pieces per operand, wave and stage");
#define TT_ISSUE(NB_)                                                          \
    do {                                                                       \
      tt_issue<(NB_), STAGES>(rsA, rsB, ldsA, ldsB, voA, voB, wv);             \
      voA[0] += stepA; voA[1] += stepA; voB[0] += stepB; voB[1] += stepB;      \
    } while (0)
#define TT_FRAGS(B_) do { tt_frags<(B_), STAGES, TNt>(faB, bfr); tt_frags<(B_), STAGES, TMt>(faA, af); } while (0)
    // LOAD slot of stage kt (ring slot B_): issue stage kt + STAGES - 1 into ring slot B_ - 1, read the fragments of stage kt, retire this
    // wave's share of stage kt + 1 (the STAGES - 2 younger stages stay in flight) and the fragment reads
#define TT_LOAD_STEADY(B_)                                 \
    do {                                                   \
      TT_ISSUE(((B_) + STAGES - 1) % STAGES);              \
      TT_FRAGS(B_);                                        \
      wait_vm<(STAGES - 2) * NDMA>();                      \
      tt_frags_ready<TMt, TNt>(af, bfr);                   \
    } while (0)
#define TT_LOAD_GENERIC(KT, B_)                                                \
    do {                                                                       \
      const bool is_ = (KT) + STAGES - 1 < nk;                                 \
      if (is_) TT_ISSUE(((B_) + STAGES - 1) % STAGES);                         \
      TT_FRAGS(B_);                                                            \
      if ((KT) + 1 < nk) {                                                     \
        int full_ = nk - 2 - (KT);                                             \
        full_ = full_ < 0 ? 0 : (full_ > STAGES - 3 ? STAGES - 3 : full_);     \
        wait_vm_even(NDMA * full_ + (is_ ? NDMA : 0));                         \
      }                                                                        \
      tt_frags_ready<TMt, TNt>(af, bfr);                                       \
    } while (0)
#define TT_MFMA() tt_mfma<TMt, TNt>(af, bfr, acc)
    // (DIAG & 32: s_memtime stamps of workgroup 0, one wave per group: period start, between its two halves, before and after the barrier -
    // tools/slot_trace_tt.py)
    uint64_t* const stamps = ((DIAG & 32) && p.U && blockIdx.x == 0 && lane == 0 && (wv & 3) == 0) ? (uint64_t*)p.U + (wv >> 2) * 8192 : nullptr;
    int sp = 0;
#define TT_ST(I) do { if ((DIAG & 32) && stamps) stamps[4 * sp + (I)] = __builtin_readcyclecounter(); } while (0)
#define TT_MID() TT_ST(1)
#define TT_BAR() do { __builtin_amdgcn_sched_barrier(0); TT_ST(2); __builtin_amdgcn_s_barrier(); TT_ST(3); ++sp; TT_ST(0); __builtin_amdgcn_sched_barrier(0); } while (0)
    if (0 < nk) TT_ISSUE(0);
    if (1 < nk) TT_ISSUE(1);
    if (2 < nk) TT_ISSUE(2);
    {
      const int younger = (nk - 1 < STAGES - 2) ? nk - 1 : STAGES - 2;
      if (younger >= 2) wait_vm<2 * NDMA>();
      else if (younger == 1) wait_vm<NDMA>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();                 // stage 0 landed for everyone
    asm volatile("" ::: "memory");
    TT_ST(0);
    const int n_steady = nk - (STAGES - 1);       // stages kt < n_steady still have a stage to issue in their LOAD slot
    if (grp == 0) {
      // period kt: LOAD kt | MFMA kt | barrier
      int kt = 0;
      for (; kt + STAGES <= n_steady; kt += STAGES) {
        TT_LOAD_STEADY(0); TT_MID(); TT_MFMA(); TT_BAR();
        TT_LOAD_STEADY(1); TT_MID(); TT_MFMA(); TT_BAR();
        TT_LOAD_STEADY(2); TT_MID(); TT_MFMA(); TT_BAR();
        TT_LOAD_STEADY(3); TT_MID(); TT_MFMA(); TT_BAR();
      }
      for (; kt < nk; kt += STAGES) {
        TT_LOAD_GENERIC(kt, 0); TT_MID(); TT_MFMA(); TT_BAR();
        if (kt + 1 < nk) { TT_LOAD_GENERIC(kt + 1, 1); TT_MID(); TT_MFMA(); TT_BAR(); }
        if (kt + 2 < nk) { TT_LOAD_GENERIC(kt + 2, 2); TT_MID(); TT_MFMA(); TT_BAR(); }
        if (kt + 3 < nk) { TT_LOAD_GENERIC(kt + 3, 3); TT_MID(); TT_MFMA(); TT_BAR(); }
      }
      TT_BAR();                                   // the period in which group 1 multiplies its last stage
    } else {
      // period 0: LOAD 0 | barrier;  period kt >= 1: MFMA kt-1 | LOAD kt | barrier;  last: MFMA nk-1 | barrier
      if (0 < nk) TT_LOAD_GENERIC(0, 0);
      TT_BAR();
      int kt = 1;
      for (; kt + STAGES <= n_steady; kt += STAGES) {       // kt = 1 (mod 4): ring slots 1, 2, 3, 0
        TT_MFMA(); TT_MID(); TT_LOAD_STEADY(1); TT_BAR();
        TT_MFMA(); TT_MID(); TT_LOAD_STEADY(2); TT_BAR();
        TT_MFMA(); TT_MID(); TT_LOAD_STEADY(3); TT_BAR();
        TT_MFMA(); TT_MID(); TT_LOAD_STEADY(0); TT_BAR();
      }
      for (; kt < nk; kt += STAGES) {
        TT_MFMA(); TT_MID(); TT_LOAD_GENERIC(kt, 1); TT_BAR();
        if (kt + 1 < nk) { TT_MFMA(); TT_MID(); TT_LOAD_GENERIC(kt + 1, 2); TT_BAR(); }
        if (kt + 2 < nk) { TT_MFMA(); TT_MID(); TT_LOAD_GENERIC(kt + 2, 3); TT_BAR(); }
        if (kt + 3 < nk) { TT_MFMA(); TT_MID(); TT_LOAD_GENERIC(kt + 3, 0); TT_BAR(); }
      }
      if (0 < nk) TT_MFMA();
      TT_BAR();
    }
#undef TT_ISSUE
#undef TT_FRAGS
#undef TT_LOAD_STEADY
#undef TT_LOAD_GENERIC
#undef TT_MFMA
#undef TT_BAR
#undef TT_MID
#undef TT_ST
  } else if constexpr (NT == 512) {
    // ---- ping-pong (8 waves = 2 groups of one wave per SIMD): a group alternates a LOAD slot (fragment
    // reads of stage j, LDS-DMA of stage j+STAGES-1, counted vmcnt) with an MFMA slot (stage j); group 1
    // runs one slot behind group 0, so on every SIMD one wave is always in its MFMA slot while the other
    // feeds itself.  One workgroup-wide s_barrier per slot keeps the two groups out of phase.
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);   // waves 0-3 / 4-7 = one wave per SIMD each; provably wave-uniform: the guarded s_barrier must be a scalar branch
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0) {
      if (s0 < nk) {
        stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + s0 * STAGE, m0a, kbeg + (int64_t)s0 * BK, lda_, tid);
        if (b_packed) stage_tile_packed<BN, NT, NB_AUX_B>(rsB, lds + s0 * STAGE + A_BYTES, (uint32_t)((tile_n * nkt + s0) * B_BYTES), tid, lds + STAGES * STAGE);
        else
        stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + s0 * STAGE + A_BYTES, n0, kbeg + (int64_t)s0 * BK, ldb_, tid, lds + STAGES * STAGE);
      }
    }
    {
      const int younger = (nk - 1 < STAGES - 2) ? nk - 1 : STAGES - 2;
      if (STAGES >= 5 && younger >= 3) wait_vm<3 * NDMA>();
      else if (younger >= 2) wait_vm<2 * NDMA>();
      else if (younger == 1) wait_vm<NDMA>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();                 // stage 0 landed for everyone
    asm volatile("" ::: "memory");
    NB_STAMP(1);
    if (grp == 1) __builtin_amdgcn_s_barrier();   // offset group 1 by one slot
    int buf = 0;
    bf16x8 af[TMt], bfr[TNt];
    for (int kt = 0; kt < nk; ++kt) {
      // ---------------- LOAD slot ----------------
      if (kt + STAGES - 1 < nk && !(DIAG & 1)) {
        int nb = buf + STAGES - 1;
        if (nb >= STAGES) nb -= STAGES;
        const int64_t k0 = kbeg + (int64_t)(kt + STAGES - 1) * BK;
        stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + nb * STAGE, m0a, k0, lda_, tid);
        if (b_packed) stage_tile_packed<BN, NT, NB_AUX_B>(rsB, lds + nb * STAGE + A_BYTES, (uint32_t)((tile_n * nkt + kt + STAGES - 1) * B_BYTES), tid, lds + STAGES * STAGE);
        else
        stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + nb * STAGE + A_BYTES, n0, k0, ldb_, tid, lds + STAGES * STAGE);
      }
      const char* cur = lds + buf * STAGE;
      if ((DIAG & 2) == 0 || kt == 0) {
#pragma unroll
        for (int j = 0; j < TNt; ++j) bfr[j] = read_frag2<TB, BN, true>(cur + A_BYTES, wn * WTN + j * 16, lane);
#pragma unroll
        for (int i = 0; i < TMt; ++i) af[i] = read_frag2<TA, BM, true>(cur, wm * WTM + i * 16, lane);
      }
      {
        // retire this wave's share of stage kt+1 (read by group 0 two slots from now)
        const int c = (nk - 1 - kt < STAGES - 1) ? nk - 1 - kt : STAGES - 1;   // stages kt+1.. outstanding
        if (STAGES >= 5 && c >= 4) wait_vm<3 * NDMA>();
        else if (c >= 3) wait_vm<2 * NDMA>();
        else if (c == 2) wait_vm<NDMA>();
        else if (c == 1) wait_vm<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((DIAG & 32) && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && p.U) ((uint64_t*)p.U)[(wave >> 2) * 8192 + 4 * kt + 0] = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();
      if ((DIAG & 32) && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && p.U) ((uint64_t*)p.U)[(wave >> 2) * 8192 + 4 * kt + 1] = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA slot ----------------
      __builtin_amdgcn_s_setprio(1);
      if constexpr (TA || TB) {   // asm reads: the compiler does not know they are pending
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // volatile asm statements keep their order: every fragment is redefined AFTER the wait, so no MFMA can be
        // scheduled above it
#pragma unroll
        for (int i = 0; i < TMt; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
        for (int j = 0; j < TNt; ++j) asm volatile("" : "+v"(bfr[j]));
      }
      if constexpr (!(DIAG & 4)) {
#pragma unroll
        for (int i = 0; i < TMt; ++i)
#pragma unroll
          for (int j = 0; j < TNt; ++j) acc[i][j] = kDirect ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < TMt; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
        for (int j = 0; j < TNt; ++j) asm volatile("" :: "v"(bfr[j]));
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if ((DIAG & 32) && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && p.U) ((uint64_t*)p.U)[(wave >> 2) * 8192 + 4 * kt + 2] = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();
      if ((DIAG & 32) && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && p.U) ((uint64_t*)p.U)[(wave >> 2) * 8192 + 4 * kt + 3] = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
      buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the barrier count
    NB_STAMP(2);
  } else if constexpr (PIPE) {
  // ---- ring: buffers hold stages kt+1 .. kt+STAGES; fragments of stage kt+1 are read (into the other
  // register set) while the MFMAs of stage kt run, so no LDS round trip is exposed at a stage boundary.
#pragma unroll
  for (int s = 0; s < STAGES; ++s) {
    if (s < nk) {
      stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + s * STAGE, m0a, kbeg + (int64_t)s * BK, lda_, tid);
      stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + s * STAGE + A_BYTES, n0, kbeg + (int64_t)s * BK, ldb_, tid);
    }
  }
  bf16x8 afA[TMt], bfA[TNt], afB[TMt], bfB[TNt];
  {
    const int inflight = (nk < STAGES ? nk : STAGES) - 1;   // stages that may stay in flight behind stage 0
    if (inflight >= 3) wait_vm<3 * NDMA>();
    else if (inflight == 2) wait_vm<2 * NDMA>();
    else if (inflight == 1) wait_vm<NDMA>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nk > 0) {
#pragma unroll
      for (int j = 0; j < TNt; ++j) bfA[j] = read_frag2<TB, BN>(lds + A_BYTES, wn * WTN + j * 16, lane);
#pragma unroll
      for (int i = 0; i < TMt; ++i) afA[i] = read_frag2<TA, BM>(lds, wm * WTM + i * 16, lane);
    }
  }
  int buf = 0;  // buffer of stage kt
  // one pipeline step: top-of-iteration sync for stage kt+1, refill the buffer stage kt occupied,
  // read fragments of stage kt+1 into (an, bn) while multiplying (ac, bc)
#define NB_STEP(ac, bc, an, bn)                                                                              \
  {                                                                                                          \
    const int c = (nk - 1 - kt < STAGES - 1) ? nk - 1 - kt : STAGES - 1; /* stages kt+1.. still outstanding */ \
    if (c >= 3) wait_vm<2 * NDMA>();                                                                         \
    else if (c == 2) wait_vm<NDMA>();                                                                        \
    else if (c == 1) wait_vm<0>();                                                                           \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("" ::: "memory");                                                                           \
    if (kt + STAGES < nk) {                                                                                  \
      const int64_t k0 = kbeg + (int64_t)(kt + STAGES) * BK;                                                 \
      stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + buf * STAGE, m0a, k0, lda_, tid);                                       \
      stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + buf * STAGE + A_BYTES, n0, k0, ldb_, tid);                             \
    }                                                                                                        \
    const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;                                                      \
    if (kt + 1 < nk) {                                                                                       \
      const char* nx = lds + nbuf * STAGE;                                                                   \
      _Pragma("unroll") for (int j = 0; j < TNt; ++j) bn[j] = read_frag2<TB, BN>(nx + A_BYTES, wn * WTN + j * 16, lane); \
      _Pragma("unroll") for (int i = 0; i < TMt; ++i) an[i] = read_frag2<TA, BM>(nx, wm * WTM + i * 16, lane); \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TMt; ++i)                                                          \
      _Pragma("unroll") for (int j = 0; j < TNt; ++j)                                                        \
        acc[i][j] = kDirect ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[i], bc[j], acc[i][j], 0, 0, 0)   \
                            : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[j], ac[i], acc[i][j], 0, 0, 0);  \
    buf = nbuf;                                                                                              \
    ++kt;                                                                                                    \
  }
  for (int kt = 0; kt < nk;) {
    NB_STEP(afA, bfA, afB, bfB)
    if (kt >= nk) break;
    NB_STEP(afB, bfB, afA, bfA)
  }
#undef NB_STEP

  } else {
  // ---- plain ring (256x128: no registers left for a second fragment set): STAGES-1 stages in flight ----
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) {
    if (s < nk) {
      stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + s * STAGE, m0a, kbeg + (int64_t)s * BK, lda_, tid);
      stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + s * STAGE + A_BYTES, n0, kbeg + (int64_t)s * BK, ldb_, tid);
    }
  }
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // retire stage kt: at most min(STAGES-2, nk-1-kt) younger stages may stay in flight
    const int younger = (nk - 1 - kt < STAGES - 2) ? nk - 1 - kt : STAGES - 2;
    if (STAGES >= 4 && younger == 2) wait_vm<2 * NDMA>();
    else if (younger >= 1) wait_vm<NDMA>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // stage kt landed for all waves; all waves finished reading stage kt-1
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < nk) {
      int nb = buf + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      const int64_t k0 = kbeg + (int64_t)(kt + STAGES - 1) * BK;
      stage_tile2<TA, BM, NT, NB_AUX_A>(rsA, lds + nb * STAGE, m0a, k0, lda_, tid);
      stage_tile2<TB, BN, NT, NB_AUX_B, kPW>(rsB, lds + nb * STAGE + A_BYTES, n0, k0, ldb_, tid, lds + STAGES * STAGE);
    }
    const char* cur = lds + buf * STAGE;
    bf16x8 af[TMt], bfr[TNt];
#pragma unroll
    for (int j = 0; j < TNt; ++j) bfr[j] = read_frag2<TB, BN>(cur + A_BYTES, wn * WTN + j * 16, lane);
#pragma unroll
    for (int i = 0; i < TMt; ++i) af[i] = read_frag2<TA, BM>(cur, wm * WTM + i * 16, lane);
#pragma unroll
    for (int i = 0; i < TMt; ++i)
#pragma unroll
      for (int j = 0; j < TNt; ++j) acc[i][j] = kDirect ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
  }
  }

  if constexpr (kDirect) {
    // ---- epilogue straight from the accumulators: no LDS, no barrier, no branch; 4 x TMt independent groups of TNt elements per
    // lane in one basic block, which the compiler is free to interleave (the GELU forms are ~25-instruction chains)
    if (kHasR || kHasUin) {
#pragma unroll
      for (int i = NPRE0; i < TMt; ++i) load_dpre(i);
    }
    float colacc[TNt];
#pragma unroll
    for (int j = 0; j < TNt; ++j) colacc[j] = 0.f;
    const uint32_t voC = (uint32_t)((drow0 * p.ldc + dcol) * 2), svC = (uint32_t)(2 * p.ldc);
    const uint32_t dbase0 = (uint32_t)(drow0 * p.N + dcol), sD = (uint32_t)p.N;
    // fused column sums: compiled into the x GELU' epilogue only (the FFN-up bias gradient - the one caller in the training step).  As a
    // run-time option of EVERY epilogue it cost each of them a select + an add per element (the compiler if-converts the guarded
    // accumulate): 256 of the 676 vector instructions of the plain-bias epilogue.  Other epilogues with colsum_out: nbest_gemm_bf16_v2_wins.
    constexpr bool kCols = (EPI == NBEST_EPI_DGELU);
    const bool want_cols = kCols && p.colpart != nullptr;
#pragma unroll
    for (int i = 0; i < TMt; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t rr = (uint32_t)(16 * i + e);
        float v[TNt];
#pragma unroll
        for (int j = 0; j < TNt; ++j) v[j] = acc[i][j][e];
        if (kHasBias) {
#pragma unroll
          for (int j = 0; j < TNt; ++j) v[j] += db[j];
        }
        if (EPI == NBEST_EPI_BIAS_GELU) {
          float gp[TNt];
#pragma unroll
          for (int j = 0; j < TNt; j += 2) {
            f32x2 h2, g2;   // gelu(u), and gelu'(u) kept for the backward
            gelu_pair_fast(f32x2{v[j], v[j + 1]}, h2, g2);
            gp[j] = g2[0]; gp[j + 1] = g2[1]; v[j] = h2[0]; v[j + 1] = h2[1];
          }
          if constexpr (TNt == 8) nb_bstore8(rsU, voU + rr * svU, gd_pack4(gp), gd_pack4(gp + 4));
          else __builtin_amdgcn_raw_buffer_store_b32(gd_pack4(gp), rsU, voU + rr * svU, 0, 2);
        }
        if (EPI == NBEST_EPI_BIAS_DROP_RES && p.drop.thr16) {
          const uint32_t dbase = dbase0 + rr * sD;
          uint32_t k = 0;       // dbase is even: one hash per element pair
#pragma unroll
          for (int q = 0; q < TNt / 2; ++q) k |= nb_keep2(p.drop, dbase + 2 * q) << (2 * q);
#pragma unroll
          for (int j = 0; j < TNt; ++j) v[j] = (k >> j & 1) ? v[j] * p.drop.scale : 0.f;
        }
        if (kHasR) {
          const bf16x8 r8 = __builtin_bit_cast(bf16x8, dpre[i][e]);
#pragma unroll
          for (int j = 0; j < TNt; ++j) v[j] += (float)r8[j];
        }
        if (kHasUin) {
          float gd[8];
          gd_unpack4(dpre[i][e][0], gd);
          if constexpr (TNt == 8) gd_unpack4(dpre[i][e][1], gd + 4);
#pragma unroll
          for (int j = 0; j < TNt; ++j) v[j] *= gd[j];
        }
        // (experiment DIAG & 4096: the QKV projection - the step's only plain-bias GEMM - written WITHOUT the streaming hint, for its
        // immediate reader, the attention forward)
        if constexpr (TNt == 8) nb_bstore_bf16x8<(EPI == NBEST_EPI_BIAS && (DIAG & 4096)) ? 0 : 2>(rsC, voC + rr * svC, v);
        else if constexpr (TNt == 6) {
          bf16x2 o0 = {(bf16)v[0], (bf16)v[1]}, o1 = {(bf16)v[2], (bf16)v[3]}, o2 = {(bf16)v[4], (bf16)v[5]};
          __builtin_amdgcn_raw_buffer_store_b96(u32x3{__builtin_bit_cast(uint32_t, o0), __builtin_bit_cast(uint32_t, o1), __builtin_bit_cast(uint32_t, o2)},
                                                rsC, voC + rr * svC, 0, 2);
        } else {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsC, voC + rr * svC, 0, 2);
        }
        if constexpr (kCols) {
          const bool live = want_cols && (drow0 + rr) < p.M;
#pragma unroll
          for (int j = 0; j < TNt; ++j) colacc[j] += live ? v[j] : 0.f;
        }
      }
    }
    if (kCols && want_cols) {   // fused bias gradient: sum over the 4 lane groups (rows 4g + e) -> one partial row per wave
#pragma unroll
      for (int j = 0; j < TNt; ++j) {
        float x = colacc[j];
        x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
        colacc[j] = x;
      }
      if (dg == 0) {
        float* o = p.colpart + ((int64_t)tile_m * WM + wm) * p.N + dcol;
#pragma unroll
        for (int q = 0; q < TNt / 2; ++q) *(f32x2*)(o + 2 * q) = f32x2{colacc[2 * q], colacc[2 * q + 1]};
      }
    }
    return;
  }
  // ---- epilogue: 32-row chunks restaged through wave-private LDS ([32][64] fp32, chunk16 ^= row & 15) ----
  float* ep = (float*)lds + wave * 2048;
  float colacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (kPre) {
#pragma unroll
    for (int c = 1; c < TMt / 2; ++c)
#pragma unroll
      for (int it = 0; it < 4; ++it) pre[c][it] = load_pre(erow0 + c * 32 + it * 8);
  }
  __builtin_amdgcn_s_barrier();   // every wave has finished reading the operand ring
  asm volatile("" ::: "memory");
  // element offsets of this lane's 8-column group in the row-major outputs, advanced by 8 rows per iteration (m * ld afresh for
  // every row = quarter-rate 64-bit multiplies in a VALU-bound epilogue)
  int64_t oC = erow0 * p.ldc + en8, oU = erow0 * p.ldu + en8;
  int64_t oS = ((int64_t)z * p.M + erow0) * p.N + en8;
  const int64_t sC = 8 * p.ldc, sU = 8 * p.ldu, sS = 8 * p.N;
  uint32_t dbase = (uint32_t)(erow0 * p.N + en8);
  const uint32_t sD = 8u * (uint32_t)p.N;
#pragma unroll
  for (int c = 0; c < TMt / 2; ++c) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int row = ii * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < TNt; ++j) {
        const int cidx = j * 4 + (lane >> 4);
        *(f32x4*)(ep + row * 64 + ((cidx ^ (row & 15)) << 2)) = acc[2 * c + ii][j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pidx = it * 64 + lane, row = pidx >> 3, c8 = pidx & 7;
      const int64_t m = m0 + wm * WTM + c * 32 + row;
      f32x4 v0 = *(const f32x4*)(ep + row * 64 + (((2 * c8) ^ (row & 15)) << 2));
      f32x4 v1 = *(const f32x4*)(ep + row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2));
      if (m < p.M) {
      if (EPI == NBEST_EPI_F32_SPLITK) {
        float* cp = (p.splits > 1) ? p.slab + oS : (float*)p.C + oC;
        if (p.splits == 1 && p.accumulate) { v0 += *(const f32x4*)cp; v1 += *(const f32x4*)(cp + 4); }
        *(f32x4*)cp = v0;
        *(f32x4*)(cp + 4) = v1;
      } else {
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (kHasBias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += pb0[e]; v[4 + e] += pb1[e]; }
      }
      if (EPI == NBEST_EPI_BIAS_GELU) {
        float gp[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2 h2, g2;   // gelu(u), and gelu'(u) kept for the backward
          gelu_pair_fast(f32x2{v[e], v[e + 1]}, h2, g2);
          gp[e] = g2[0]; gp[e + 1] = g2[1]; v[e] = h2[0]; v[e + 1] = h2[1];
        }
        st_stream((i32x2*)((uint8_t*)p.U + oU), i32x2{(int)gd_pack4(gp), (int)gd_pack4(gp + 4)}, p.stream_out);
      }
      if (EPI == NBEST_EPI_BIAS_DROP_RES && p.drop.thr16) {
        const uint32_t k = nb_keep4(p.drop, dbase) | (nb_keep4(p.drop, dbase + 4) << 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (k >> e & 1) ? v[e] * p.drop.scale : 0.f;
      }
      if (kHasR) {
        const bf16x8 r = __builtin_bit_cast(bf16x8, pre[c][it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
      }
      if (kHasUin) {
        float gd[8];
        gd_unpack4((uint32_t)pre[c][it][0], gd);
        gd_unpack4((uint32_t)pre[c][it][1], gd + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gd[e];
      }
      st_stream_bf16x8((bf16*)p.C + oC, v, p.stream_out);
      if (EPI != NBEST_EPI_F32_SPLITK && p.colpart) {
#pragma unroll
        for (int e = 0; e < 8; ++e) colacc[e] += v[e];
      }
      }
      }
      oC += sC; oU += sU; oS += sS; dbase += sD;
    }
    asm volatile("" ::: "memory");
  }
  if (DIAG & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stamp after the stores are acknowledged
  NB_STAMP(3);
  if (EPI != NBEST_EPI_F32_SPLITK && p.colpart) {   // fused bias gradient: per-wave column sums -> partial rows
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = colacc[e];
      x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
      colacc[e] = x;
    }
    if ((lane >> 3) == 0) {
      float* o = p.colpart + ((int64_t)tile_m * WM + wm) * p.N + en8;
      *(f32x4*)o = f32x4{colacc[0], colacc[1], colacc[2], colacc[3]};
      *(f32x4*)(o + 4) = f32x4{colacc[4], colacc[5], colacc[6], colacc[7]};
    }
  }
}

#ifdef NBEST_EXPERIMENTS
// LDS / global accesses that the wait-count pass must not see (it would drain the in-flight LDS-DMA ring, and every
// store with it, in front of each of them); the caller orders them with explicit s_waitcnt.
__device__ __forceinline__ void lds_write_b128_asm(const void* addr, f32x4 v) {
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)addr;
  asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 lds_read_b128_asm(const void* addr) {
  f32x4 v;
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)addr;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 global_load_f32x4_asm(const float* ptr) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}

// ---- persistent ping-pong kernel -------------------------------------------------------------------------------
// One workgroup per CU walks tiles b, b+G, b+2G, ...; the 4-stage LDS ring simply runs on into the next tile: the
// last three LOAD slots of a tile issue stages 0..2 of the NEXT tile, so its prologue (≈10 % of a K = 768 tile,
// the LDS-DMA latency of the first stage) is hidden behind the current tile's MFMAs and epilogue.  While those
// three stages sit in three ring buffers the epilogue restages through the fourth (32 KiB: 4 KiB per wave, 16-row
// chunks).  k-contiguous operands, epilogues without a residual / GELU' input (NONE, BIAS, BIAS_GELU), splits == 1,
// nk >= 3.  vmcnt: the epilogue's stores (and the tile's bias loads) enter the per-wave stream between DMA stages;
// completion is in order, so the counted waits of the first two LOAD slots of a tile allow for them explicitly
// (full tiles issue a fixed number of stores; the ragged last row tile drains with vmcnt(0) instead).
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm2p_kernel(GemmP2 p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int BM = 256, BN = 256, WN = 4, NT = 512, STAGES = 4;
  constexpr int WTM = 128, WTN = 64, TMt = 8, TNt = 4;
  constexpr int A_BYTES = BM * BK * 2, STAGE = 2 * A_BYTES;
  constexpr int NDMA = (BM + BN) * 4 / NT;   // 4
  constexpr bool kDirect = false;            // LDS-restaged epilogue, swapped MFMA operands
  constexpr int kPW = 0;
  constexpr bool kHasBias = (EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU);
  constexpr int NSTORE = (WTM / 8) * (EPI == NBEST_EPI_BIAS_GELU ? 2 : 1);   // store instructions per wave and (full) tile
  constexpr int NBIAS = kHasBias ? 2 : 0;
  static_assert(EPI == NBEST_EPI_NONE || EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU, "no residual / GELU' input");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int grp = __builtin_amdgcn_readfirstlane(wm);
  const int G = gridDim.x, tiles = p.tiles_m * p.tiles_n;
  const int nk = (int)(p.K / BK);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  auto tile_origin = [&](int t, int64_t& m0, int64_t& n0) {
    const int id = xcd_remap2(t, tiles);
    const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
    m0 = (int64_t)tile_m * BM; n0 = (int64_t)tile_n * BN;
  };
  auto issue = [&](int gstage, int64_t m0, int64_t n0, int s) {   // stage s of the tile at (m0, n0) -> ring buffer gstage & 3
    char* dst = lds + (gstage & 3) * STAGE;
    stage_tile2<false, BM, NT, NB_AUX_A>(rsA, dst, m0, (int64_t)s * BK, p.lda, tid);
    stage_tile2<false, BN, NT, NB_AUX_B>(rsB, dst + A_BYTES, n0, (int64_t)s * BK, p.ldb, tid);
  };

  int t_cur = blockIdx.x;
  int64_t m0, n0;
  tile_origin(t_cur, m0, n0);
  int gs = 0;   // ring position of stage 0 of the current tile
#pragma unroll
  for (int s = 0; s < 3; ++s) issue(gs + s, m0, n0, s);
  int dbg_i = 0;
#define NB_PSTAMP() if ((DIAG & 32) && EPI == NBEST_EPI_NONE && p.U && tid == 0 && blockIdx.x == 0 && dbg_i < 64) ((uint64_t*)p.U)[dbg_i++] = __builtin_readcyclecounter()
  NB_PSTAMP();
  if ((DIAG & 32) && EPI == NBEST_EPI_NONE && p.U && tid == 0) ((uint64_t*)p.U)[1000 + 2 * blockIdx.x] = __builtin_readcyclecounter();
  int extra = 0;   // younger non-DMA operations (stores of the previous tile, bias loads) that may precede stage 3 in the stream

  while (true) {
    const int t_next = t_cur + G;
    const bool has_next = t_next < tiles;
    int64_t m0n = 0, n0n = 0;
    if (has_next) tile_origin(t_next, m0n, n0n);
    const int64_t en8 = n0 + wn * WTN + (lane & 7) * 8;
    f32x4 pb0 = {0, 0, 0, 0}, pb1 = {0, 0, 0, 0};
    if (kHasBias) { pb0 = global_load_f32x4_asm(p.bias + en8); pb1 = global_load_f32x4_asm(p.bias + en8 + 4); }
    f32x4 acc[TMt][TNt];
#pragma unroll
    for (int i = 0; i < TMt; ++i)
#pragma unroll
      for (int j = 0; j < TNt; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // stage 0 landed: stream tail is  s0 s1 s2 [stores of the previous tile] [bias loads]
    if (extra < 0) wait_vm<0>();
    else if (extra == 0) wait_vm<2 * NDMA + NBIAS>();   // first tile of this workgroup: no stores in the stream yet
    else wait_vm<2 * NDMA + NSTORE + NBIAS>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    NB_PSTAMP();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // offset group 1 by one slot
    bf16x8 af[TMt], bfr[TNt];
    for (int kt = 0; kt < nk; ++kt) {
      // ---------------- LOAD slot ----------------
      const int si = kt + 3;
      if (si < nk) issue(gs + si, m0, n0, si);
      else if (has_next) issue(gs + si, m0n, n0n, si - nk);
      const char* cur = lds + ((gs + kt) & 3) * STAGE;
#pragma unroll
      for (int j = 0; j < TNt; ++j) bfr[j] = read_frag2<false, BN>(cur + A_BYTES, wn * WTN + j * 16, lane);
#pragma unroll
      for (int i = 0; i < TMt; ++i) af[i] = read_frag2<false, BM>(cur, wm * WTM + i * 16, lane);
      {
        // stage kt+1 must have landed; younger: the stages issued after it (+ at kt = 0, 1 the non-DMA operations
        // that sit between stage 2 and stage 3 of this tile in the stream)
        const int c = has_next ? 3 : ((nk - 1 - kt < 3) ? nk - 1 - kt : 3);   // stages kt+1 .. outstanding
        if (c >= 3) {
          if (kt >= 2) wait_vm<2 * NDMA>();
          else if (extra < 0) wait_vm<0>();
          else if (extra == 0) wait_vm<2 * NDMA + NBIAS>();
          else wait_vm<2 * NDMA + NSTORE + NBIAS>();
        } else if (c == 2) wait_vm<NDMA>();   // (with nk < 5 this over-waits for the mixed operations at kt < 2: safe)
        else wait_vm<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA slot ----------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TMt; ++i)
#pragma unroll
        for (int j = 0; j < TNt; ++j) acc[i][j] = kDirect ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the barrier count: both groups are in step again
    NB_PSTAMP();

    // ---- epilogue: 16-row chunks through the one ring buffer the next tile's first three stages do not occupy ----
    float* ep = (float*)(lds + ((gs + nk + 3) & 3) * STAGE) + wave * 1024;
    const bool full = (m0 + BM <= p.M);
    // the bias loads of this tile are older than its stage 3, and stage nk-1 has landed (in-order completion)
    if (kHasBias) asm volatile("" : "+v"(pb0), "+v"(pb1));
#pragma unroll
    for (int c = 0; c < TMt; ++c) {
      const int wrow = lane & 15;
#pragma unroll
      for (int j = 0; j < TNt; ++j) {
        const int cidx = j * 4 + (lane >> 4);
        lds_write_b128_asm(ep + wrow * 64 + ((cidx ^ wrow) << 2), acc[c][j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 rv[2][2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int pidx = it * 64 + lane, row = pidx >> 3, c8 = pidx & 7;
        rv[it][0] = lds_read_b128_asm(ep + row * 64 + (((2 * c8) ^ row) << 2));
        rv[it][1] = lds_read_b128_asm(ep + row * 64 + (((2 * c8 + 1) ^ row) << 2));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(rv[0][0]), "+v"(rv[0][1]), "+v"(rv[1][0]), "+v"(rv[1][1]));
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int pidx = it * 64 + lane, row = pidx >> 3;
        const int64_t m = m0 + wm * WTM + c * 16 + row;
        const f32x4 v0 = rv[it][0], v1 = rv[it][1];
        if (m >= p.M) continue;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (kHasBias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += pb0[e]; v[4 + e] += pb1[e]; }
        }
        if (EPI == NBEST_EPI_BIAS_GELU) {
          float gp[8];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            f32x2 h2, g2;   // gelu(u), and gelu'(u) kept for the backward
            gelu_pair_fast(f32x2{v[e], v[e + 1]}, h2, g2);
            gp[e] = g2[0]; gp[e + 1] = g2[1]; v[e] = h2[0]; v[e + 1] = h2[1];
          }
          st_stream((i32x2*)((uint8_t*)p.U + m * p.ldu + en8), i32x2{(int)gd_pack4(gp), (int)gd_pack4(gp + 4)}, p.stream_out);
        }
        st_stream_bf16x8((bf16*)p.C + m * p.ldc + en8, v, p.stream_out);
      }
      asm volatile("" ::: "memory");
    }
    NB_PSTAMP();
    if ((DIAG & 32) && EPI == NBEST_EPI_NONE && p.U && tid == 0 && !has_next) ((uint64_t*)p.U)[1001 + 2 * blockIdx.x] = __builtin_readcyclecounter();
    if (!has_next) break;
    __builtin_amdgcn_s_barrier();   // every wave is done with the restage buffer: stage 3 of the next tile may overwrite it
    asm volatile("" ::: "memory");
    extra = full ? NSTORE : -1;      // ragged tile: unknown store count -> the next tile drains with vmcnt(0)
    gs += nk;
    t_cur = t_next; m0 = m0n; n0 = n0n;
  }
}


// ---- persistent ping-pong kernel with the REGISTER epilogue (experiment, NBEST_PERSISTENT=2): one workgroup per CU walks tiles
// t = blockIdx, blockIdx + G, ...; the LDS ring runs on into the next tile (its first three stages are issued during the last three
// LOAD slots of the current one), so neither the LDS-DMA latency of a tile's prologue nor the workgroup launch is exposed, and the
// epilogue - registers and buffer stores only, no LDS - overlaps the landing of those stages.  k-contiguous operands, 256 x 256 tiles,
// 4 x 2 waves of 64 x 128, epilogues NONE / BIAS / BIAS_GELU.  vmcnt bookkeeping: per tile and wave the stream is
//   S0 S1 S2 (issued inside the previous tile) | NST epilogue stores of the previous tile | 2 bias loads | S3 S4 ...
// and operations complete in order, so "stage kt+1 landed" allows 2 stages + NST + 2 in flight at kt < 2 and 2 stages afterwards.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm2d_kernel(GemmP2 p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int BM = 256, BN = 256, WN = 2, NT = 512, STAGES = 4;
  constexpr int WTM = 64, WTN = 128, TMt = 4, TNt = 8;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NDMA = 4;
  constexpr bool kHasBias = (EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU);
  constexpr int NST = TMt * 4 * (EPI == NBEST_EPI_BIAS_GELU ? 2 : 1);   // store wave-instructions per tile and wave (ragged tiles too: range-checked)
  constexpr int NBIAS = kHasBias ? 2 : 0;
  static_assert(EPI == NBEST_EPI_NONE || EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU, "no residual / GELU' input");
  static_assert(2 * NDMA + NST + NBIAS <= 63, "vmcnt is a 6-bit counter");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  const int G = gridDim.x, tiles = p.tiles_m * p.tiles_n;
  const int nk = (int)(p.K / BK);
  const bool b_packed = p.Bp != nullptr;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(b_packed ? (void*)p.Bp : (void*)p.B, 0, b_packed ? p.bp_bytes : p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, p.u_bytes, 0x00020000);
  auto tile_coords = [&](int t, int& tm, int& tn) {
    const int id = xcd_remap2(t, tiles);
    nb_tile_coords(id, p.tiles_m, p.gn, tm, tn);
  };
  auto issue = [&](int gstage, int tm, int tn, int s) {   // stage s of tile (tm, tn) -> ring buffer gstage & 3
    char* dst = lds + (gstage & 3) * STAGE;
    stage_tile2<false, BM, NT, NB_AUX_A>(rsA, dst, (int64_t)tm * BM, (int64_t)s * BK, p.lda, tid);
    if (b_packed) stage_tile_packed<BN, NT, NB_AUX_B>(rsB, dst + A_BYTES, (uint32_t)((tn * nk + s) * B_BYTES), tid);
    else stage_tile2<false, BN, NT, NB_AUX_B, WTN>(rsB, dst + A_BYTES, (int64_t)tn * BN, (int64_t)s * BK, p.ldb, tid);
  };
  const int dg = lane >> 4, dc = lane & 15;
  int t_cur = blockIdx.x, tm, tn;
  tile_coords(t_cur, tm, tn);
  int gs = 0;   // ring position of stage 0 of the current tile
#pragma unroll
  for (int s = 0; s < 3; ++s) issue(gs + s, tm, tn, s);
  bool first = true;
  while (true) {
    const int t_next = t_cur + G;
    const bool has_next = t_next < tiles;
    int tmn = 0, tnn = 0;
    if (has_next) tile_coords(t_next, tmn, tnn);
    const int64_t dcol = (int64_t)tn * BN + wn * WTN + TNt * dc;
    const int64_t drow0 = (int64_t)tm * BM + wm * WTM + 4 * dg;
    f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    if (kHasBias) { b0 = global_load_f32x4_asm(p.bias + dcol); b1 = global_load_f32x4_asm(p.bias + dcol + 4); }
    f32x4 acc[TMt][TNt];
#pragma unroll
    for (int i = 0; i < TMt; ++i)
#pragma unroll
      for (int j = 0; j < TNt; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    if (first) wait_vm<2 * NDMA + NBIAS>();
    else wait_vm<2 * NDMA + NST + NBIAS>();
    __builtin_amdgcn_s_barrier();                 // stage 0 of this tile landed for everyone
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();   // offset group 1 by one slot
    bf16x8 af[TMt], bfr[TNt];
    for (int kt = 0; kt < nk; ++kt) {
      // ---------------- LOAD slot ----------------
      const int si = kt + 3;
      if (si < nk) issue(gs + si, tm, tn, si);
      else if (has_next) issue(gs + si, tmn, tnn, si - nk);
      const char* cur = lds + ((gs + kt) & 3) * STAGE;
#pragma unroll
      for (int j = 0; j < TNt; ++j) bfr[j] = read_frag2<false, BN>(cur + A_BYTES, wn * WTN + j * 16, lane);
#pragma unroll
      for (int i = 0; i < TMt; ++i) af[i] = read_frag2<false, BM>(cur, wm * WTM + i * 16, lane);
      {
        const int c = has_next ? 3 : ((nk - 1 - kt < 3) ? nk - 1 - kt : 3);   // stages kt+1 .. outstanding
        if (c >= 3) {
          if (kt >= 2) wait_vm<2 * NDMA>();
          else if (first) wait_vm<2 * NDMA + NBIAS>();
          else wait_vm<2 * NDMA + NST + NBIAS>();
        } else if (c == 2) wait_vm<NDMA>();    // (tail of the last tile; over-waits at kt < 2 when nk < 5: safe)
        else wait_vm<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA slot ----------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TMt; ++i)
#pragma unroll
        for (int j = 0; j < TNt; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the barrier count: both groups are in step again
    // ---- register epilogue (gemm2_kernel's, reduced to these epilogues): bias loads are older than stage 3, which has landed ----
    if (kHasBias) asm volatile("" : "+v"(b0), "+v"(b1));
    const float db[TNt] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    const uint32_t voC = (uint32_t)((drow0 * p.ldc + dcol) * 2), svC = (uint32_t)(2 * p.ldc);
    const uint32_t voU = (uint32_t)(drow0 * p.ldu + dcol), svU = (uint32_t)p.ldu;
#pragma unroll
    for (int i = 0; i < TMt; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t rr = (uint32_t)(16 * i + e);
        float v[TNt];
#pragma unroll
        for (int j = 0; j < TNt; ++j) v[j] = acc[i][j][e] + (kHasBias ? db[j] : 0.f);
        if (EPI == NBEST_EPI_BIAS_GELU) {
          float gp[TNt];
#pragma unroll
          for (int j = 0; j < TNt; j += 2) {
            f32x2 h2, g2;
            gelu_pair_fast(f32x2{v[j], v[j + 1]}, h2, g2);
            gp[j] = g2[0]; gp[j + 1] = g2[1]; v[j] = h2[0]; v[j + 1] = h2[1];
          }
          nb_bstore8(rsU, voU + rr * svU, gd_pack4(gp), gd_pack4(gp + 4));
        }
        nb_bstore_bf16x8(rsC, voC + rr * svC, v);
      }
    }
    if (!has_next) break;
    t_cur = t_next; tm = tmn; tn = tnn; gs += nk; first = false;
  }
}

#endif  // NBEST_EXPERIMENTS

// rows >= m_split of the slabs' [M][N] image belong to the second output of a weight-gradient pair (C2, ldc2); m_split = M: none
__global__ __launch_bounds__(256) void splitk_reduce2_kernel(const float* __restrict__ slab, float* __restrict__ C, int64_t MN,
                                                             int64_t N, int64_t ldc, int splits, int accumulate,
                                                             float* __restrict__ C2, int64_t m_split, int64_t ldc2) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < MN; i += (int64_t)gridDim.x * blockDim.x * 4) {
    f32x4 s = *(const f32x4*)(slab + i);
    for (int z = 1; z < splits; ++z) s += *(const f32x4*)(slab + (int64_t)z * MN + i);
    const int64_t m = i / N, n = i - m * N;
    float* c = (m < m_split) ? C + m * ldc + n : C2 + (m - m_split) * ldc2 + n;
    if (accumulate) s += *(const f32x4*)c;
    *(f32x4*)c = s;
  }
}

// B operand (a weight matrix [N][K], k-contiguous) -> the order the 256 x bn ping-pong kernel stages it: for every tile column and
// K stage the bn x 32 LDS image (chunk swizzle and the register epilogue's row permutation applied), contiguous.  LDS-DMA with
// 64-byte row segments delivers 22 B/clk/CU, contiguous 47 (tools/micro/dma_rate.hip); with MFMAs removed the k-contiguous GEMMs
// still take 75-85 % of their time (DMA + barriers), so the B half of the staging traffic is worth packing once per optimizer step.
__global__ __launch_bounds__(256) void pack_b_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                     const nbest_matrix_desc* __restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const nbest_matrix_desc d = descs[lo];
  const int bn = d.pad, pw = (bn == 256) ? 128 : 96;            // wave-tile width of the kernel that reads it (kPW)
  const int nk = d.cols / BK, t = blockIdx.x - d.tile_start, tile_n = t / nk, kt = t - tile_n * nk;
  const bf16* s = src + d.offset + ((int64_t)tile_n * bn) * d.cols + (int64_t)kt * BK;
  bf16* o = dst + d.offset + ((int64_t)tile_n * nk + kt) * bn * BK;
  for (int p = threadIdx.x; p < bn * 4; p += blockDim.x) {
    const int row = p >> 2, slot = p & 3;
    const int kc = slot ^ ((-(row >> 2)) & 3);                   // stage_tile2<false>: the chunk swizzle
    const int x = row % pw;
    const int grow = row - x + (pw / 16) * (x & 15) + (x >> 4);  // and its PW row permutation
    *(i32x4*)(o + (int64_t)p * 8) = *(const i32x4*)(s + (int64_t)grow * d.cols + kc * 8);
  }
}

struct Plan {
  int bm, bn, splits;
  int64_t kps;
};

// The shipped library has ONE code path per shape and reads no environment.  `make diag` (-DNBEST_EXPERIMENTS) builds
// csrc/diag/libnbest_diag.so, in which NBEST_PERSISTENT=1 selects the persistent kernel and NBEST_TILE=256x256 | 256x128 |
// 128x256 | 128x128 forces a tile (tools/ load it through NBEST_LIB).
#ifdef NBEST_EXPERIMENTS
static bool persistent_enabled() {
  static const bool v = [] { const char* e = getenv("NBEST_PERSISTENT"); return e && *e == '1'; }();
  return v;
}
static bool persistent_direct_enabled() {
  static const bool v = [] { const char* e = getenv("NBEST_PERSISTENT"); return e && *e == '2'; }();
  return v;
}
static bool stages5_enabled() {
  static const bool v = [] { const char* e = getenv("NBEST_STAGES"); return e && *e == '5'; }();
  return v;
}
static bool sym_enabled() {
  static const bool v = [] { const char* e = getenv("NBEST_SYM"); return e && *e == '1'; }();
  return v;
}
static int forced_tile() {
  static const int v = [] {
    const char* e = getenv("NBEST_TILE");
    if (!e) return 0;
    if (!strcmp(e, "256x256")) return 3;
    if (!strcmp(e, "256x192")) return 5;
    if (!strcmp(e, "128x256")) return 4;
    if (!strcmp(e, "256x128")) return 2;
    if (!strcmp(e, "128x128")) return 1;
    if (!strcmp(e, "128x384")) return 6;
    if (!strcmp(e, "128x512")) return 7;
    return 0;
  }();
  return v;
}
#else
static constexpr bool persistent_enabled() { return false; }
static constexpr bool sym_enabled() { return false; }
static constexpr bool stages5_enabled() { return false; }
static constexpr int forced_tile() { return 0; }
#endif

static Plan make_plan(const nbest_gemm_args* a) {
  Plan pl;
  pl.bn = 128;
  // 256x128 where it fills the chip (two workgroups per CU -> 512 slots) and is not a weight gradient
  const int64_t t256 = ((a->M + 255) / 256) * (a->N / 128);
  pl.bm = (!a->trans_a && t256 >= 1024) ? 256 : 128;
  const int ft6 = forced_tile();
  const int ft = (ft6 == 6 || ft6 == 7) ? 0 : ft6;      // 128x384 is an ADDITIONAL choice for the N = 768 shapes: every other shape plans as usual
  const bool ok256 = (a->N % 256 == 0) && (!a->trans_a || a->M % 256 == 0);
  if (ft == 3 && ok256) { pl.bm = 256; pl.bn = 256; }
  else if (ft == 4 && ok256) { pl.bm = 128; pl.bn = 256; }
  else if (ft == 2 && !a->trans_a) { pl.bm = 256; pl.bn = 128; }
  else if (ft == 1) { pl.bm = 128; pl.bn = 128; }
  else if (ft == 0 && !a->trans_a && !a->trans_b && a->N % 256 == 0 &&
           [&] {   // >= 4 rounds of tiles on the 256 CUs, or at least one round with the last one >= 85 % full
             const int64_t t = ((a->M + 255) / 256) * (a->N / 256);
             return t >= 1024 || (t >= 256 && (double)t / (double)(((t + 255) / 256) * 256) >= 0.85);
           }()) {
    pl.bm = 256; pl.bn = 256;   // ping-pong schedule: best for k-contiguous operands on the wide GEMMs (QKV, FFN-up forward); round 3: also
                                // for 1 - 3 full rounds (xlm-roberta-large, M = 16 384: 1 478 -> 1 562 utt/s with every N % 256 == 0 GEMM on it)
  } else if (ft == 0 && a->trans_a && a->trans_b && a->epilogue == NBEST_EPI_F32_SPLITK && ok256 &&
             (a->M / 256) * (a->N / 256) >= 18) {
    pl.bm = 256; pl.bn = 256;   // weight gradients with >= 18 output tiles (QKV, FFN): 1.0-1.05 PFLOP/s vs 0.85-0.94 for v1;
                                // the 768x768 attention-output gradient (9 tiles, 28 splits) stays on v1 (0.90 vs 0.79)
  }
  // 256 x 192 tiles (ping-pong, 4 x 2 waves of 64 x 96 columns): N = 768 gives 4 x (M / 256) tiles - 512 = exactly two rounds on the
  // 256 CUs at M = 32 768, where 256 x 256 tiles give 384 = one and a half (a quarter of the chip idle for half the kernel) and
  // the 128 x 128 kernel three rounds of a structure that tops out near 1 PFLOP/s.  Chosen from one full round of tiles when its rounds are fuller (with a
  // handicap for the smaller tile: 24 instead of 32 MFMAs per barrier pair); plain / bias / residual epilogues only.
  // Measured in the step: N = 768 (256-wide tiles: 75 % full rounds) - on a par with the 128 x 128 kernel it replaces (112 / 148 us
  // for the forward / dgrad pairs against 114 / 146), 4-6 % faster in isolation; N = 2304 (90 % full rounds) - 127 us against 116-121
  // for 256 x 256: the handicap is 15 %, so only shapes whose 256-wide rounds are under 85 % full take it.
  {
    const bool epi192 = a->epilogue == NBEST_EPI_NONE || a->epilogue == NBEST_EPI_BIAS || a->epilogue == NBEST_EPI_BIAS_DROP_RES ||
                        a->epilogue == NBEST_EPI_RES;
    if (!a->trans_a && !a->trans_b && a->N % 192 == 0 && epi192 && (ft == 0 || ft == 5)) {
      const int64_t rows = (a->M + 255) / 256, t192 = rows * (a->N / 192);
      auto eff = [](int64_t t) { return (double)t / (double)(((t + 255) / 256) * 256); };
      const double e256 = (a->N % 256 == 0) ? eff(rows * (a->N / 256)) : 0.0;
      const double ecur = (pl.bm == 256 && pl.bn == 256) ? e256 : (a->N % 256 == 0 ? e256 : 0.0);
      int64_t t192_min = 256;   // one full round of tiles is enough (was two): M = 16 384 rows - configs[3] +2.6 %, a 128-utterance batch +4.8 %
#ifdef NBEST_EXPERIMENTS
      if (const char* e = getenv("NBEST_T192MIN")) t192_min = atoll(e);
#endif
      double handicap = 0.85;
#ifdef NBEST_EXPERIMENTS
      if (const char* e = getenv("NBEST_T192H")) handicap = atof(e);
#endif
      if (ft == 5 || (t192 >= t192_min && handicap * eff(t192) > ecur)) { pl.bm = 256; pl.bn = 192; }
    }
    // 128 x 384 tiles (2 x 4 waves of 64 x 96; round 4): the same MFMA work per stage as 256 x 192 with HALF the bytes of the A operand - the
    // activation panel, which in the training step comes cold from HBM - and twice those of the weight panel, which the L2
    // holds (and which arrives packed: two adjacent 192-column blocks).  Takes over wherever 256 x 192 was chosen and N is a multiple of
    // 384.  Same call, alternating: the five N = 768 GEMMs of a layer 620 -> 586 us cold / 541 -> 510 warm; the step 21.15 / 21.22 -> 20.85 / 20.88 ms.
    if (pl.bm == 256 && pl.bn == 192 && a->N % 384 == 0 && a->K % BK == 0 && ft != 5) { pl.bm = 128; pl.bn = 384; }
    // 128 x 512 (2 x 4 waves of 64 x 128, 40 KB stages: all 160 KB of LDS): the same turn for the N = 1024 shapes of xlm-roberta-large, which plan
    // 256 x 256 (4 tile columns): xlm-roberta-large S = 256, 64 utterances 1 585 -> 1 622 utt/s (same call, twice).  NBEST_TILE=256x256 keeps the old plan.
    if (ft6 != 3 && pl.bm == 256 && pl.bn == 256 && !a->trans_a && !a->trans_b && epi192 && a->N % 512 == 0 && a->N <= 2048 && a->K % BK == 0) { pl.bm = 128; pl.bn = 512; }
#ifdef NBEST_EXPERIMENTS
    {   // NBEST_T512ALL=1: every 256 x 256 plan with N % 512 == 0, GELU epilogues included
      static const int all = [] { const char* e = getenv("NBEST_T512ALL"); return (e && e[0] == '1') ? 1 : 0; }();
      if (all && pl.bm == 256 && pl.bn == 256 && !a->trans_a && !a->trans_b && a->epilogue != NBEST_EPI_F32_SPLITK && a->N % 512 == 0 && a->K % BK == 0) { pl.bm = 128; pl.bn = 512; }
    }
#endif
  }
  const int64_t tiles = ((a->M + pl.bm - 1) / pl.bm) * (a->N / pl.bn);
  const int64_t slots = (pl.bn >= 192) ? 256 : 512;   // workgroups resident at once
  int64_t splits = 1;
  if (a->epilogue == NBEST_EPI_F32_SPLITK) {
    const int64_t maxs = (a->K / 512 < 1) ? 1 : ((a->K / 512 > 32) ? 32 : a->K / 512);
    double best = -1.0;
    for (int64_t sp = 1; sp <= maxs; ++sp) {
      const int64_t blocks = tiles * sp;
      const double eff = (double)blocks / (double)(((blocks + slots - 1) / slots) * slots);
      if (eff > best + 1e-9) { best = eff; splits = sp; }
      if (blocks * 5 >= slots * 4 && eff >= 0.93) { splits = sp; break; }
    }
  }
  int64_t k = (a->K + splits - 1) / splits;
  k = (k + 63) / 64 * 64;
  pl.splits = (int)((a->K + k - 1) / k);
  pl.kps = k;
  return pl;
}

template <int BM, int BN, int WM, int WN, int STAGES, bool TA, bool TB, bool SYM = false>
static int launch2(const GemmP2& p, int epi, int grid, hipStream_t st) {
  constexpr int NT = WM * WN * 64;
  constexpr int lds_ring = STAGES * (BM + BN) * BK * 2;
  constexpr int lds_pairs = (!TA && !TB && NT == 512 && BM == 256 && (DIAG & 2048)) ? (3 * 2 * BM + STAGES * BN) * BK * 2 : 0;   // A staged in pairs of stages (3 pair buffers)
  constexpr int lds_bytes = (lds_pairs > lds_ring ? lds_pairs : lds_ring) + (((BN * 4) % NT) ? (NT / 64) * 1024 : 0);   // + the zero-fill dump slots
#define L(E)                                                                                                        \
  case E:                                                                                                           \
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<BM, BN, WM, WN, STAGES, TA, TB, E, SYM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
    gemm2_kernel<BM, BN, WM, WN, STAGES, TA, TB, E, SYM><<<grid, NT, lds_bytes, st>>>(p);                            \
    break;
  if constexpr (BN / WN == 64) {   // the fp32 split-K output goes through the LDS-restaged epilogue (64-column wave tiles)
    if (epi == NBEST_EPI_F32_SPLITK) {
      switch (epi) { L(NBEST_EPI_F32_SPLITK) }
      NB_LAUNCH_CHECK();
      return NBEST_OK;
    }
  }
  if constexpr (BN / WN == 96) {   // 96-column wave tiles: no 8-bit GELU' rows
    switch (epi) {
      L(NBEST_EPI_NONE) L(NBEST_EPI_BIAS) L(NBEST_EPI_BIAS_DROP_RES) L(NBEST_EPI_RES)
      default:
        nbest_set_error("gemm: epilogue %d is not built for 192-column tiles", epi);
        return NBEST_ERR_ARG;
    }
  } else {
    switch (epi) {
      L(NBEST_EPI_NONE) L(NBEST_EPI_BIAS) L(NBEST_EPI_BIAS_GELU) L(NBEST_EPI_BIAS_DROP_RES) L(NBEST_EPI_DGELU)
      L(NBEST_EPI_RES)
      default:
        nbest_set_error("gemm: bad epilogue %d", epi);
        return NBEST_ERR_ARG;
    }
  }
#undef L
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

}  // namespace

// v2 is only faster with its 256x128 tile (measured: +8..14 % on the wide-N forward / dgrad GEMMs); the
// 128x128 BK=32 ring loses to v1's 128x128 BK=64 (half the MFMAs per barrier and per DMA instruction)
bool nbest_gemm_bf16_v2_wins(const nbest_gemm_args* a) {
  // column sums fused into an epilogue other than x GELU': the generation-1 kernel carries them for every epilogue (no caller in the training step)
  if (a->colsum_out && a->epilogue != NBEST_EPI_DGELU && a->epilogue != NBEST_EPI_F32_SPLITK && !a->trans_a && !a->trans_b) return false;
  const Plan pl = make_plan(a);
  return pl.bm == 256 || (pl.bm == 128 && pl.bn >= 384) || (forced_tile() != 0 && forced_tile() < 6);
}   // 256x128 ring or 256x256 ping-pong

int nbest_internal_partial_rows_sum(const float* part, int nrows, int N, float* out, int accumulate, hipStream_t st);

size_t nbest_gemm_bf16_v2_ws_bytes(const nbest_gemm_args* a) {
  if (a->epilogue != NBEST_EPI_F32_SPLITK) {
    if (!a->colsum_out) return 0;
    const Plan pl0 = make_plan(a);
    return (size_t)((a->M + pl0.bm - 1) / pl0.bm) * 4 * a->N * sizeof(float);   // one partial row per wave row of a tile (<= 4)
  }
  const Plan pl = make_plan(a);
  return pl.splits > 1 ? (size_t)pl.splits * a->M * a->N * sizeof(float) : 0;
}

// `b2` (with `a` = the VIRTUAL problem of M1 + M2 output rows carrying the first problem's pointers, m_split = M1): the second
// problem of a weight-gradient pair, see nbest_wgrad_pair_bf16
static int gemm_v2_impl(const nbest_gemm_args* a, const nbest_gemm_args* b2, int64_t m_split, hipStream_t st) {
  NB_CHECK(a->N % 64 == 0, NBEST_ERR_SHAPE, "gemm(bf16): N=%lld must be a multiple of 64", (long long)a->N);
  NB_CHECK(a->trans_a || a->K % BK == 0, NBEST_ERR_SHAPE, "gemm(bf16): K=%lld must be a multiple of %d", (long long)a->K, BK);
  NB_CHECK(!(a->trans_a && !a->trans_b), NBEST_ERR_ARG, "gemm(bf16): trans_a without trans_b is not built");
  NB_CHECK(!a->trans_a || a->M % 128 == 0, NBEST_ERR_SHAPE, "gemm(bf16): trans_a needs M %% 128 == 0");
  NB_CHECK(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 8 == 0, NBEST_ERR_ALIGN, "gemm(bf16): leading dimensions must be multiples of 8");
  NB_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0 && ((uintptr_t)a->C & 15) == 0, NBEST_ERR_ALIGN,
           "gemm(bf16): pointers must be 16-byte aligned");
  const Plan pl = make_plan(a);
  GemmP2 p;
  p.A = (const bf16*)a->A; p.B = (const bf16*)a->B; p.C = a->C; p.bias = a->bias; p.R = (const bf16*)a->R; p.U = (bf16*)a->U;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldu = a->ldu;
  p.tiles_m = (int)((a->M + pl.bm - 1) / pl.bm);
  p.tiles_n = (int)(a->N / pl.bn);
  p.splits = pl.splits;
  p.k_per_split = pl.kps;
  p.accumulate = a->accumulate;
  p.slab = (float*)a->ws;
  p.colpart = nullptr;
  p.A2 = p.B2 = nullptr; p.lda2 = p.ldb2 = p.m_split = 0; p.a2_bytes = p.b2_bytes = 0;
  p.Bp = nullptr; p.bp_bytes = 0; p.bp_bn = 0;
  if (a->B_packed && ((a->b_pack_bn == pl.bn && pl.bm == 256 && (pl.bn == 256 || pl.bn == 192)) || (pl.bm == 128 && pl.bn == 384 && a->b_pack_bn == 192) || (pl.bm == 128 && pl.bn == 512 && a->b_pack_bn == 256)) &&
      !a->trans_a && !a->trans_b &&
      a->epilogue != NBEST_EPI_F32_SPLITK && a->K % BK == 0 && a->N * a->K * 2 < ((int64_t)1 << 32) && ((uintptr_t)a->B_packed & 15) == 0) {
    p.Bp = (const bf16*)a->B_packed;
    p.bp_bytes = (uint32_t)(a->N * a->K * 2);
    p.bp_bn = a->b_pack_bn;
  }
  if (a->colsum_out && a->epilogue != NBEST_EPI_F32_SPLITK) {
    NB_CHECK(a->epilogue == NBEST_EPI_DGELU || a->trans_a || a->trans_b, NBEST_ERR_ARG,
             "gemm(bf16, generation 2): column sums are fused into the x GELU' epilogue only (nbest_gemm routes the others to generation 1)");
    NB_CHECK(a->ws && a->ws_bytes >= nbest_gemm_bf16_v2_ws_bytes(a), NBEST_ERR_WORKSPACE, "gemm: column-sum workspace too small");
    p.colpart = (float*)a->ws;
  }
  const int64_t a_rows = a->trans_a ? a->K : a->M, a_cols = a->trans_a ? (b2 ? m_split : a->M) : a->K;
  const int64_t b_rows = a->trans_b ? a->K : a->N, b_cols = a->trans_b ? a->N : a->K;
  const int64_t ab = ((a_rows - 1) * a->lda + a_cols) * 2, bb = ((b_rows - 1) * a->ldb + b_cols) * 2;
  NB_CHECK(ab < ((int64_t)1 << 32) && bb < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "gemm(bf16): operand larger than 4 GiB");
  p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  if (b2) {
    const int64_t ab2 = ((a->K - 1) * b2->lda + b2->M) * 2, bb2 = ((a->K - 1) * b2->ldb + a->N) * 2;
    NB_CHECK(ab2 < ((int64_t)1 << 32) && bb2 < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "gemm(bf16): operand larger than 4 GiB");
    p.A2 = (const bf16*)b2->A; p.B2 = (const bf16*)b2->B; p.lda2 = b2->lda; p.ldb2 = b2->ldb; p.m_split = m_split;
    p.a2_bytes = (uint32_t)ab2; p.b2_bytes = (uint32_t)bb2;
    NB_CHECK(pl.bm == 256 && pl.bn == 256 && pl.splits > 1 && m_split % 256 == 0, NBEST_ERR_SHAPE, "wgrad pair: not a 256 x 256 split-K plan");
  }
  p.c_bytes = p.r_bytes = p.u_bytes = 0;
  if (a->epilogue != NBEST_EPI_F32_SPLITK) {   // 32-bit byte offsets, also for the rows of a ragged last tile (dropped by the range check)
    const int64_t mpad = a->M + 256;
    NB_CHECK(mpad * a->ldc * 2 < ((int64_t)1 << 32) && mpad * a->ldr * 2 < ((int64_t)1 << 32) && mpad * a->ldu < ((int64_t)1 << 32),
             NBEST_ERR_SHAPE, "gemm(bf16): output / residual larger than 4 GiB");
    p.c_bytes = (uint32_t)(a->M * a->ldc * 2);
    p.r_bytes = a->R ? (uint32_t)(a->M * a->ldr * 2) : 0;
    p.u_bytes = a->U ? (uint32_t)(a->M * a->ldu) : 0;
  }
  p.drop = make_drop(a->drop_p, a->seed, a->drop_stream);
  p.stream_out = nb_stream_output(a->M * a->N * 2) ? 1 : 0;
  // B is a weight matrix (k-contiguous [N][K]) in the forward / dgrad GEMMs; the weight gradients have no small operand
  // B is a weight matrix (k-contiguous [N][K]) in the forward / dgrad GEMMs; the weight gradients have no small operand.
  // (Column groups re-read the ACTIVATION panel once per group: 490 MB of HBM-side traffic per launch for 277 MB of operands on the
  // N = 768 dgrads, profiles/r03_pmc.csv - the re-read panel was written by the previous kernel and comes out of the Infinity
  // Cache.  Same-box sweep of the five N = 768 launches of a layer: 606 / 584 / 578 us with 1 / 2 / 4 tile columns per group - one
  // column per group loses 4 %, two (the rule's choice at K >= 2304) and row-major tie.)
  p.gn = (!a->trans_a && !a->trans_b) ? nb_group_cols(a->N / pl.bn, (int64_t)pl.bn * a->K * 2, 2400) : (int)(a->N / pl.bn);
  // 128 x 384 tiles: the two tile columns of an M row run back to back on one XCD, so the (cold) activation panel crosses the fabric once -
  // the weight panels of two columns (2 x 2.4 MB at K = 3072) exceed the 2 400 KB slice rule above, but the workgroups of an XCD walk K
  // in step and only a few stages of them are live at a time.  Same call: the five N = 768 GEMMs of a layer 587 / 600 -> 570 / 576 us cold,
  // the step 21.02 / 21.01 -> 20.86 / 20.85 ms.
  if (pl.bm == 128 && pl.bn >= 384 && (a->N / pl.bn) % 2 == 0) p.gn = 2;
#ifdef NBEST_EXPERIMENTS
  if (pl.bn == 384) { if (const char* e = getenv("NBEST_GN384")) { const int v = atoi(e); if (v > 0 && (a->N / pl.bn) % v == 0) p.gn = v; } }
#endif
  NB_CHECK(a->M * a->N < ((int64_t)1 << 32) || p.drop.thr16 == 0, NBEST_ERR_SHAPE, "gemm(bf16): dropout counter overflow");
  const int epi = a->epilogue;
  if (epi == NBEST_EPI_BIAS || epi == NBEST_EPI_BIAS_GELU || epi == NBEST_EPI_BIAS_DROP_RES)
    NB_CHECK(a->bias, NBEST_ERR_ARG, "gemm: epilogue %d needs bias", epi);
  if (epi == NBEST_EPI_BIAS_DROP_RES || epi == NBEST_EPI_RES)
    NB_CHECK(a->R && a->ldr % 8 == 0 && ((uintptr_t)a->R & 15) == 0, NBEST_ERR_ARG, "gemm: epilogue %d needs R", epi);
  if (epi == NBEST_EPI_BIAS_GELU || epi == NBEST_EPI_DGELU)   // U: 8-bit GELU' rows (gd_pack4), ldu in bytes
    NB_CHECK(a->U && a->ldu % 8 == 0 && ((uintptr_t)a->U & 15) == 0, NBEST_ERR_ARG, "gemm: epilogue %d needs U", epi);
  if (epi == NBEST_EPI_F32_SPLITK && p.splits > 1)
    NB_CHECK(a->ws && a->ws_bytes >= (size_t)p.splits * a->M * a->N * sizeof(float), NBEST_ERR_WORKSPACE,
             "gemm: split-K workspace too small (%zu < %zu)", a->ws_bytes, (size_t)p.splits * a->M * a->N * sizeof(float));
  const int grid = p.tiles_m * p.tiles_n * p.splits;
  int rc, wave_rows = 2;   // wave rows per tile = partial rows of the fused column sums
  NB_CHECK(a->N % pl.bn == 0, NBEST_ERR_SHAPE, "gemm(bf16): N=%lld is not a multiple of the %d-column tile", (long long)a->N, pl.bn);
  if (pl.bm == 128 && pl.bn == 512) {
    rc = launch2<128, 512, 2, 4, 4, false, false>(p, epi, grid, st);
  } else if (pl.bm == 128 && pl.bn == 384) {
    if (a->K >= 2048 || stages5_enabled()) rc = launch2<128, 384, 2, 4, 5, false, false>(p, epi, grid, st);
    else rc = launch2<128, 384, 2, 4, 4, false, false>(p, epi, grid, st);
  } else if (pl.bm == 256 && pl.bn == 192) {
    // ring depth: the operand delivery of these kernels is bound by bytes in flight against the LDS-DMA latency (3 stages of 28-32 KB
    // against ~2 us); a fifth stage (all 160 KB of LDS at 256 x 256) pays at long K - FFN-down forward 162 -> 155 us, FFN-up dgrad 157 ->
    // 154, QKV dgrad 122 -> 120 - and costs 1-2 % at K = 768, where the longer prologue of each tile weighs more (same-call A/B, twice)
    if (a->K >= 2048 || stages5_enabled()) rc = launch2<256, 192, 4, 2, 5, false, false>(p, epi, grid, st);
    else rc = launch2<256, 192, 4, 2, 4, false, false>(p, epi, grid, st);
    wave_rows = 4;
  } else if (pl.bm == 128 && pl.bn == 256) {
    if (!a->trans_a && !a->trans_b) rc = launch2<128, 256, 2, 4, 4, false, false>(p, epi, grid, st);
    else if (!a->trans_a && a->trans_b) rc = launch2<128, 256, 2, 4, 4, false, true>(p, epi, grid, st);
    else rc = launch2<128, 256, 2, 4, 4, true, true>(p, epi, grid, st);
#ifdef NBEST_EXPERIMENTS
  } else if (pl.bm == 256 && pl.bn == 256 && !a->trans_a && !a->trans_b && pl.splits == 1 && !p.colpart && a->K % BK == 0 &&
             a->K >= 3 * BK && grid > 256 && (epi == NBEST_EPI_NONE || epi == NBEST_EPI_BIAS || epi == NBEST_EPI_BIAS_GELU) &&
             persistent_enabled()) {
    constexpr int lds_bytes = 4 * 2 * 256 * BK * 2;
    const int pgrid = 256;
#define LP(E)                                                                                              \
  case E:                                                                                                  \
    (void)hipFuncSetAttribute((const void*)gemm2p_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
    gemm2p_kernel<E><<<pgrid, 512, lds_bytes, st>>>(p);                                                    \
    break;
    switch (epi) { LP(NBEST_EPI_NONE) LP(NBEST_EPI_BIAS) LP(NBEST_EPI_BIAS_GELU) default: break; }
#undef LP
    NB_LAUNCH_CHECK();
    rc = NBEST_OK;
#endif
#ifdef NBEST_EXPERIMENTS
  } else if (pl.bm == 256 && pl.bn == 256 && !a->trans_a && !a->trans_b && pl.splits == 1 && !p.colpart && a->K % BK == 0 &&
             a->K >= 5 * BK && grid > 256 && (epi == NBEST_EPI_NONE || epi == NBEST_EPI_BIAS || epi == NBEST_EPI_BIAS_GELU) &&
             persistent_direct_enabled()) {
    constexpr int lds_bytes = 4 * 2 * 256 * BK * 2;
#define LD(E)                                                                                              \
  case E:                                                                                                  \
    (void)hipFuncSetAttribute((const void*)gemm2d_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
    gemm2d_kernel<E><<<256, 512, lds_bytes, st>>>(p);                                                      \
    break;
    switch (epi) { LD(NBEST_EPI_NONE) LD(NBEST_EPI_BIAS) LD(NBEST_EPI_BIAS_GELU) default: break; }
#undef LD
    NB_LAUNCH_CHECK();
    rc = NBEST_OK; wave_rows = 4;
#endif
  } else if (pl.bm == 256 && pl.bn == 256) {
    // k-contiguous operands: 4 x 2 waves with 64 x 128 wave tiles (register epilogue: 16-byte stores, whole 128-byte lines)
    if (!a->trans_a && !a->trans_b && epi != NBEST_EPI_F32_SPLITK) {
#ifdef NBEST_EXPERIMENTS
      if (sym_enabled()) rc = launch2<256, 256, 4, 2, 4, false, false, true>(p, epi, grid, st);
      else
#endif
      if (a->K >= 2048 || stages5_enabled()) rc = launch2<256, 256, 4, 2, 5, false, false>(p, epi, grid, st);
      else rc = launch2<256, 256, 4, 2, 4, false, false>(p, epi, grid, st);
      wave_rows = 4;
    }
    else if (!a->trans_a && !a->trans_b) rc = launch2<256, 256, 2, 4, 4, false, false>(p, epi, grid, st);
    else if (!a->trans_a && a->trans_b) rc = launch2<256, 256, 2, 4, 4, false, true>(p, epi, grid, st);
    else if (stages5_enabled()) rc = launch2<256, 256, 2, 4, 5, true, true>(p, epi, grid, st);
    else rc = launch2<256, 256, 2, 4, 4, true, true>(p, epi, grid, st);
  } else if (pl.bm == 256) {
    if (!a->trans_b && epi != NBEST_EPI_F32_SPLITK) { rc = launch2<256, 128, 4, 1, 3, false, false>(p, epi, grid, st); wave_rows = 4; }
    else if (!a->trans_b) rc = launch2<256, 128, 2, 2, 3, false, false>(p, epi, grid, st);
    else rc = launch2<256, 128, 2, 2, 3, false, true>(p, epi, grid, st);
  } else {
    if (!a->trans_a && !a->trans_b) rc = launch2<128, 128, 2, 2, 4, false, false>(p, epi, grid, st);
    else if (!a->trans_a && a->trans_b) rc = launch2<128, 128, 2, 2, 4, false, true>(p, epi, grid, st);
    else rc = launch2<128, 128, 2, 2, 4, true, true>(p, epi, grid, st);
  }
  if (rc) return rc;
  if (p.colpart) return nbest_internal_partial_rows_sum(p.colpart, p.tiles_m * wave_rows, (int)a->N, a->colsum_out, a->colsum_accumulate, st);
  if (epi == NBEST_EPI_F32_SPLITK && p.splits > 1 && !(a->flags & NBEST_GEMM_DEFER_REDUCE)) {
    const int64_t MN = a->M * a->N;
    int64_t g = (MN / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    splitk_reduce2_kernel<<<(int)g, 256, 0, st>>>(p.slab, (float*)a->C, MN, a->N, a->ldc, p.splits, a->accumulate,
                                                  b2 ? (float*)b2->C : nullptr, b2 ? m_split : a->M, b2 ? b2->ldc : 0);
    NB_LAUNCH_CHECK();
  }
  return NBEST_OK;
}

int nbest_gemm_bf16_v2(const nbest_gemm_args* a, hipStream_t st) { return gemm_v2_impl(a, nullptr, 0, st); }

// Two weight gradients with the same K (token rows) and N in ONE launch: the output tiles of the second are appended below the
// first's (a virtual [M1 + M2][N] result; its slabs are reduced into the two gradients by one reduce launch).  The 768 x 768
// attention-output gradient (9 tiles of 256 x 256: 28 K-splits on its own) rides with the 2304 x 768 QKV gradient (27 tiles):
// 36 tiles x 7 splits = the FFN gradients' launch shape.  Returns NBEST_ERR_SHAPE when the pair does not fit the 256 x 256 plan
// (the caller then issues the two GEMMs separately).
static bool pair_virtual(const nbest_gemm_args* a, const nbest_gemm_args* b, nbest_gemm_args* v) {
  if (a->dtype != NBEST_BF16 || b->dtype != NBEST_BF16 || !a->trans_a || !a->trans_b || !b->trans_a || !b->trans_b) return false;
  if (a->epilogue != NBEST_EPI_F32_SPLITK || b->epilogue != NBEST_EPI_F32_SPLITK) return false;
  if (a->N != b->N || a->K != b->K || a->accumulate != b->accumulate || a->M % 256 || b->M % 256 || a->N % 256) return false;
  if ((a->flags | b->flags) & NBEST_GEMM_DEFER_REDUCE) return false;
  *v = *a;
  v->M = a->M + b->M;
  const Plan pl = make_plan(v);
  return pl.bm == 256 && pl.bn == 256 && pl.splits > 1;
}
size_t nbest_wgrad_pair_bf16_ws_bytes(const nbest_gemm_args* a, const nbest_gemm_args* b) {
  nbest_gemm_args v;
  if (!pair_virtual(a, b, &v)) return 0;
  return nbest_gemm_bf16_v2_ws_bytes(&v);
}
int nbest_wgrad_pair_bf16(const nbest_gemm_args* a, const nbest_gemm_args* b, hipStream_t st) {
  nbest_gemm_args v;
  NB_CHECK(pair_virtual(a, b, &v), NBEST_ERR_SHAPE, "wgrad pair: the two problems do not share one 256 x 256 split-K launch");
  NB_CHECK(b->lda % 8 == 0 && b->ldb % 8 == 0 && b->ldc % 8 == 0 && ((uintptr_t)b->A & 15) == 0 && ((uintptr_t)b->B & 15) == 0 &&
               ((uintptr_t)b->C & 15) == 0, NBEST_ERR_ALIGN, "wgrad pair: second problem misaligned");
  return gemm_v2_impl(&v, b, a->M, st);
}

// tile width the k-contiguous GEMM of an [N][K] weight matrix is packed for (0: not packed): the rule of make_plan at training-size
// token counts - 256-column tiles for the wide matrices, 192-column tiles for N = 768-like shapes
int nbest_pack_bn_internal(int64_t N) {
  if (N % 256 == 0 && N >= 1024) return 256;
  if (N % 192 == 0) return 192;
  return 0;
}

int nbest_pack_weights_bf16(const void* src, void* dst, const nbest_matrix_desc* descs, int n_matrices, int n_stages, hipStream_t st) {
  pack_b_kernel<<<n_stages, 256, 0, st>>>((const bf16*)src, (bf16*)dst, descs, n_matrices);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}
