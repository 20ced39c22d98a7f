// K2/K4/K6: bf16 GEMM on MFMA (v_mfma_f32_16x16x32_bf16) for gfx950, fp32 accumulate, fused epilogues.
//
//   C[M,N] = epi( op(A)[M,K] . op(B)[K,N] )       block tile 128 x 128 x 64, 4 waves (2 x 2), wave tile 64 x 64
//
// HBM -> LDS : buffer_load_dwordx4 ... lds (LDS-DMA, 16 B/lane, no VGPR round trip).  The buffer
//              descriptor's range check returns ZERO for rows past the end of a matrix, which is how
//              ragged token counts (M, or K of a weight gradient) are handled without branches.
// LDS image  : LDS-DMA writes lane-linear, so the bank-conflict swizzle is applied to the per-lane
//              SOURCE address and to the read address (same involution on both sides):
//                k-contiguous operand  [128 rows][64 k]  128-B rows : chunk16 ^= (row>>1)&7   -> ds_read_b128
//                k-strided operand     [64 k][128 cols]  256-B rows : chunk16 ^= 2*(k&3)+8*((k>>3)&1) -> ds_read_b64_tr_b16
//              (transposed read = the hardware 4x16 transpose, so A^T / B^T operands of the backward
//              GEMMs need no transposed copy of activations or weights in HBM).
// pipeline   : 2 LDS stages; the DMA of tile t+1 is in flight while tile t is multiplied; one barrier per tile.
// MFMA       : operands are passed SWAPPED (B fragment first) so each lane ends up with 4 CONSECUTIVE
//              output columns of one row -> 8-byte bf16 / 16-byte fp32 epilogue accesses, bias as one float4.
// grid       : 1-D, XCD-aware bijective remap so the n-tiles that share an A row panel run on one XCD (one L2).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;       // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;   // A + B
constexpr int kLdsBytes = 2 * kStageBytes;    // 2 stages = 64 KiB

struct GemmP {
  const bf16* A; const bf16* B; void* C; const float* bias; const bf16* R; bf16* U; float* slab; float* colpart;
  int64_t M, N, K, lda, ldb, ldc, ldr, ldu;
  int64_t k_per_split;
  int tiles_m, tiles_n, splits, accumulate;
  uint32_t a_bytes, b_bytes;
  DropCfg drop;
  int stream_out;     // epilogue stores are nontemporal (common.h st_stream)
  int gn;             // tile columns per L2 group (common.h nb_tile_coords)
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- staging: one operand tile, 4 LDS-DMA instructions per thread -------------------------------
template <bool TR, int AUX = 0>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, char* lds_tile, int64_t row0, int64_t k0, int64_t ld,
                                           int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = i * 256 + tid;  // linear 16-B chunk index inside the tile
    uint32_t voff;
    if (!TR) {
      const int row = p >> 3, slot = p & 7;
      const int kc = slot ^ ((row >> 1) & 7);
      voff = (uint32_t)(((row0 + row) * ld + k0 + kc * 8) * 2);
    } else {
      const int krow = p >> 4, slot = p & 15;
      const int mc = slot ^ (2 * (krow & 3) + 8 * ((krow >> 3) & 1));
      voff = (uint32_t)(((k0 + krow) * ld + row0 + mc * 8) * 2);
    }
    char* dst = lds_tile + (i * 256 + wave * 64) * 16;  // wave-uniform; the DMA adds lane*16
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, voff, 0, 0, AUX);
  }
}

// ---- fragment reads ------------------------------------------------------------------------------
// 16 rows x 32 k fragment for v_mfma_f32_16x16x32_bf16: lane l holds row (l&15), k = 8*(l>>4) + j
template <bool TR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row_base, int ks, int lane) {
  if (!TR) {
    const int row = row_base + (lane & 15);
    const int kc = (lane >> 4) + 4 * ks;
    return *(const bf16x8*)(lds_tile + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    const int col = row_base + 4 * pq;
    bf16x8 out;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int krow = 32 * ks + 8 * g + 4 * half + q;
      const int f = 2 * (krow & 3) + 8 * ((krow >> 3) & 1);
      const char* addr = lds_tile + krow * 256 + (((col >> 3) ^ f) << 4) + ((col & 4) ? 8 : 0);
      bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)addr);
      out[4 * half + 0] = v[0]; out[4 * half + 1] = v[1]; out[4 * half + 2] = v[2]; out[4 * half + 3] = v[3];
    }
    return out;
  }
}

template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tiles = p.tiles_m * p.tiles_n;
  const int z = id / tiles, t = id - z * tiles;
  int tile_m, tile_n;
  nb_tile_coords(t, p.tiles_m, p.gn, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
  const int64_t kbeg = (int64_t)z * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nk = (int)((kend - kbeg + BK - 1) / BK);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  // epilogue operands are independent of the K loop: fetch them now so their latency hides under it
  // (the row-contiguous epilogue layout: iteration `it` -> row it*8 + lane/8, 8 columns at (lane&7)*8)
  constexpr bool kHasBias = (EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU || EPI == NBEST_EPI_BIAS_DROP_RES);
  constexpr bool kHasR = (EPI == NBEST_EPI_BIAS_DROP_RES || EPI == NBEST_EPI_RES);
  constexpr bool kHasUin = (EPI == NBEST_EPI_DGELU);
  const int64_t en8 = n0 + wn * 64 + (lane & 7) * 8;
  const int64_t erow0 = m0 + wm * 64 + (lane >> 3);
  f32x4 pb0 = {0, 0, 0, 0}, pb1 = {0, 0, 0, 0};
  if (kHasBias) { pb0 = *(const f32x4*)(p.bias + en8); pb1 = *(const f32x4*)(p.bias + en8 + 4); }
  i32x4 pre[8];
  if (kHasR) {              // residual rows: bf16, 16 bytes per lane
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t m = erow0 + it * 8;
      pre[it] = (m < p.M) ? *(const i32x4*)(p.R + m * p.ldr + en8) : i32x4{0, 0, 0, 0};
    }
  }
  if (kHasUin) {            // GELU' rows: 8-bit (gd_pack4), 8 bytes per lane, kept in .x/.y
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t m = erow0 + it * 8;
      const i32x2 q = (m < p.M) ? *(const i32x2*)((const uint8_t*)p.U + m * p.ldu + en8) : i32x2{0, 0};
      pre[it] = i32x4{q[0], q[1], 0, 0};
    }
  }

  if (nk > 0) {
    stage_tile<TA, NB_AUX_A>(rsA, lds, m0, kbeg, p.lda, tid);
    stage_tile<TB, NB_AUX_B>(rsB, lds + kTileBytes, n0, kbeg, p.ldb, tid);
  }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for every wave; every wave is done reading the other stage
    const char* cur = lds + (kt & 1) * kStageBytes;
    if constexpr (TA || TB) {
      // Transposed operands: the compiler cannot prove that a ds_read_b64_tr_b16 does not alias an LDS-DMA still in
      // flight and puts `s_waitcnt vmcnt(0)` in front of the first transposed read that follows a DMA issue in
      // program order - which used to drain tile kt+1 before tile kt was multiplied (no load/compute overlap at
      // all inside a workgroup).  So: read EVERY fragment of tile kt first, issue the DMA of tile kt+1 after the
      // last read, then multiply.  64 fragment registers; the accumulators live in AGPRs.
      bf16x8 af[2][4], bfr[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = read_frag<TA>(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[ks][j] = read_frag<TB>(cur + kTileBytes, wn * 64 + j * 16, ks, lane);
      }
      if (kt + 1 < nk) {
        char* nxt = lds + ((kt + 1) & 1) * kStageBytes;
        const int64_t k0 = kbeg + (int64_t)(kt + 1) * BK;
        stage_tile<TA, NB_AUX_A>(rsA, nxt, m0, k0, p.lda, tid);
        stage_tile<TB, NB_AUX_B>(rsB, nxt + kTileBytes, n0, k0, p.ldb, tid);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    } else {
      if (kt + 1 < nk) {
        char* nxt = lds + ((kt + 1) & 1) * kStageBytes;
        const int64_t k0 = kbeg + (int64_t)(kt + 1) * BK;
        stage_tile<TA, NB_AUX_A>(rsA, nxt, m0, k0, p.lda, tid);
        stage_tile<TB, NB_AUX_B>(rsB, nxt + kTileBytes, n0, k0, p.ldb, tid);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = read_frag<TA>(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB>(cur + kTileBytes, wn * 64 + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------
  // The accumulators (lane = 4 consecutive columns of one row, 16 rows per instruction) are restaged
  // through LDS into a row-contiguous layout: each lane then owns 8 consecutive columns of one row, so
  // residual / pre-activation loads and all stores are 16-byte accesses covering whole 128-byte lines
  // (8-byte partial-line stores made the epilogue store-issue bound).  fp32 staging: bias, GELU,
  // dropout and residual are applied in fp32 before the single rounding to bf16.
  // Wave-private region: [64 rows][64 cols] fp32 = 16 KiB, 256-B rows, 16-B chunk index ^= row & 15.
  __syncthreads();  // every wave has finished reading the operand stages
  float* ep = (float*)lds + wave * 4096;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = j * 4 + (lane >> 4);
      *(f32x4*)(ep + row * 64 + ((c ^ (row & 15)) << 2)) = acc[i][j];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float colacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // element offsets of this lane's 8-column group in the row-major outputs, advanced by 8 rows per iteration
  // (m * ld afresh for every row = quarter-rate 64-bit multiplies in a VALU-bound epilogue)
  const int64_t er0 = m0 + wm * 64 + (lane >> 3), ec8 = n0 + wn * 64 + (lane & 7) * 8;
  int64_t oC = er0 * p.ldc + ec8, oU = er0 * p.ldu + ec8, oS = ((int64_t)z * p.M + er0) * p.N + ec8;
  const int64_t sC = 8 * p.ldc, sU = 8 * p.ldu, sS = 8 * p.N;
  uint32_t dbase = (uint32_t)(er0 * p.N + ec8);
  const uint32_t sD = 8u * (uint32_t)p.N;
#pragma unroll
  for (int it = 0; it < 8; ++it, oC += sC, oU += sU, oS += sS, dbase += sD) {
    const int pidx = it * 64 + lane, row = pidx >> 3, c8 = pidx & 7;
    const int64_t m = m0 + wm * 64 + row;
    f32x4 v0 = *(const f32x4*)(ep + row * 64 + (((2 * c8) ^ (row & 15)) << 2));
    f32x4 v1 = *(const f32x4*)(ep + row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2));
    if (m >= p.M) continue;
    if (EPI == NBEST_EPI_F32_SPLITK) {
      float* c = (p.splits > 1) ? p.slab + oS : (float*)p.C + oC;
      if (p.splits == 1 && p.accumulate) { v0 += *(const f32x4*)c; v1 += *(const f32x4*)(c + 4); }
      *(f32x4*)c = v0;
      *(f32x4*)(c + 4) = v1;
      continue;
    }
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU || EPI == NBEST_EPI_BIAS_DROP_RES) {
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
    if (EPI == NBEST_EPI_BIAS_DROP_RES) {
      if (p.drop.thr16) {
        const uint32_t k = nb_keep4(p.drop, dbase) | (nb_keep4(p.drop, dbase + 4) << 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (k >> e & 1) ? v[e] * p.drop.scale : 0.f;
      }
    }
    if (EPI == NBEST_EPI_BIAS_DROP_RES || EPI == NBEST_EPI_RES) {
      const bf16x8 r = __builtin_bit_cast(bf16x8, pre[it]);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
    }
    if (EPI == NBEST_EPI_DGELU) {
      float gd[8];
      gd_unpack4((uint32_t)pre[it][0], gd);
      gd_unpack4((uint32_t)pre[it][1], gd + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gd[e];
    }
    st_stream_bf16x8((bf16*)p.C + oC, v, p.stream_out);
    if (EPI != NBEST_EPI_F32_SPLITK && p.colpart) {
#pragma unroll
      for (int e = 0; e < 8; ++e) colacc[e] += v[e];
    }
  }
  if (EPI != NBEST_EPI_F32_SPLITK && p.colpart) {   // fused bias gradient: per-wave column sums -> partial rows
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = colacc[e];
      x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
      colacc[e] = x;
    }
    if ((lane >> 3) == 0) {
      float* o = p.colpart + ((int64_t)tile_m * 2 + wm) * p.N + n0 + wn * 64 + (lane & 7) * 8;
      *(f32x4*)o = f32x4{colacc[0], colacc[1], colacc[2], colacc[3]};
      *(f32x4*)(o + 4) = f32x4{colacc[4], colacc[5], colacc[6], colacc[7]};
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int64_t MN,
                                                            int64_t N, int64_t ldc, int splits, int accumulate) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < MN; i += (int64_t)gridDim.x * blockDim.x * 4) {
    f32x4 s = *(const f32x4*)(slab + i);
    for (int z = 1; z < splits; ++z) s += *(const f32x4*)(slab + (int64_t)z * MN + i);
    const int64_t m = i / N, n = i - m * N;
    float* c = C + m * ldc + n;
    if (accumulate) s += *(const f32x4*)c;
    *(f32x4*)c = s;
  }
}

static int choose_splits(const nbest_gemm_args* a, int64_t* kps) {
  const int64_t tiles = ((a->M + BM - 1) / BM) * (a->N / BN);
  int64_t splits = 1;
  if (a->epilogue == NBEST_EPI_F32_SPLITK) {
    // the chip runs 512 workgroups at a time (2 per CU): pick the smallest split count whose grid
    // fills whole rounds (>= 93 %), so no round runs half empty; each split keeps K >= 512
    const int64_t maxs = (a->K / 512 < 1) ? 1 : ((a->K / 512 > 32) ? 32 : a->K / 512);
    double best = -1.0;
    for (int64_t s = 1; s <= maxs; ++s) {
      const int64_t blocks = tiles * s;
      const double eff = (double)blocks / (double)(((blocks + 511) / 512) * 512);
      if (eff > best + 1e-9) { best = eff; splits = s; }
      if (blocks >= 400 && eff >= 0.93) { splits = s; break; }
    }
  }
  int64_t k = (a->K + splits - 1) / splits;
  k = (k + BK - 1) / BK * BK;
  splits = (a->K + k - 1) / k;
  *kps = k;
  return (int)splits;
}

template <bool TA, bool TB>
static int launch_epi(const GemmP& p, int epi, int grid, hipStream_t st) {
#define L(E)                                                                          \
  case E:                                                                             \
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<TA, TB, E>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes); \
    gemm_bf16_kernel<TA, TB, E><<<grid, 256, kLdsBytes, st>>>(p);                      \
    break;
  switch (epi) {
    L(NBEST_EPI_NONE) L(NBEST_EPI_BIAS) L(NBEST_EPI_BIAS_GELU) L(NBEST_EPI_BIAS_DROP_RES) L(NBEST_EPI_DGELU)
    L(NBEST_EPI_RES) L(NBEST_EPI_F32_SPLITK)
    default:
      nbest_set_error("gemm: bad epilogue %d", epi);
      return NBEST_ERR_ARG;
  }
#undef L
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

}  // namespace

int nbest_internal_partial_rows_sum(const float* part, int nrows, int N, float* out, int accumulate, hipStream_t st);

size_t nbest_gemm_bf16_ws_bytes(const nbest_gemm_args* a) {
  if (a->epilogue != NBEST_EPI_F32_SPLITK) return a->colsum_out ? (size_t)((a->M + BM - 1) / BM) * 2 * a->N * sizeof(float) : 0;
  int64_t kps;
  const int splits = choose_splits(a, &kps);
  return splits > 1 ? (size_t)splits * a->M * a->N * sizeof(float) : 0;
}

int nbest_gemm_bf16(const nbest_gemm_args* a, hipStream_t st) {
  NB_CHECK(a->N % BN == 0, NBEST_ERR_SHAPE, "gemm(bf16): N=%lld must be a multiple of %d", (long long)a->N, BN);
  NB_CHECK(a->trans_a || a->K % BK == 0, NBEST_ERR_SHAPE, "gemm(bf16): K=%lld must be a multiple of %d", (long long)a->K, BK);
  NB_CHECK(!(a->trans_a && !a->trans_b), NBEST_ERR_ARG, "gemm(bf16): trans_a without trans_b is not built");
  NB_CHECK(!a->trans_a || a->M % BM == 0, NBEST_ERR_SHAPE, "gemm(bf16): trans_a needs M %% 128 == 0");
  NB_CHECK(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 8 == 0, NBEST_ERR_ALIGN, "gemm(bf16): leading dimensions must be multiples of 8");
  NB_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0 && ((uintptr_t)a->C & 15) == 0, NBEST_ERR_ALIGN,
           "gemm(bf16): pointers must be 16-byte aligned");
  GemmP p;
  p.A = (const bf16*)a->A; p.B = (const bf16*)a->B; p.C = a->C; p.bias = a->bias; p.R = (const bf16*)a->R; p.U = (bf16*)a->U;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldu = a->ldu;
  p.tiles_m = (int)((a->M + BM - 1) / BM);
  p.tiles_n = (int)(a->N / BN);
  p.splits = choose_splits(a, &p.k_per_split);
  p.accumulate = a->accumulate;
  p.slab = (float*)a->ws;
  p.colpart = nullptr;
  if (a->colsum_out && a->epilogue != NBEST_EPI_F32_SPLITK) {
    NB_CHECK(a->ws && a->ws_bytes >= nbest_gemm_bf16_ws_bytes(a), NBEST_ERR_WORKSPACE, "gemm: column-sum workspace too small");
    p.colpart = (float*)a->ws;
  }
  const int64_t a_rows = a->trans_a ? a->K : a->M, a_cols = a->trans_a ? a->M : a->K;
  const int64_t b_rows = a->trans_b ? a->K : a->N, b_cols = a->trans_b ? a->N : a->K;
  const int64_t ab = ((a_rows - 1) * a->lda + a_cols) * 2, bb = ((b_rows - 1) * a->ldb + b_cols) * 2;
  NB_CHECK(ab < ((int64_t)1 << 32) && bb < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "gemm(bf16): operand larger than 4 GiB");
  p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  p.drop = make_drop(a->drop_p, a->seed, a->drop_stream);
  p.stream_out = nb_stream_output(a->M * a->N * 2) ? 1 : 0;
  p.gn = (int)(a->N / BN);      // row-major tile order (column groups measured neutral to negative on the N = 768 shapes this kernel serves)
  NB_CHECK(a->M * a->N < ((int64_t)1 << 32) || p.drop.thr16 == 0, NBEST_ERR_SHAPE, "gemm(bf16): dropout counter overflow");
  const int epi = a->epilogue;
  if (epi == NBEST_EPI_BIAS || epi == NBEST_EPI_BIAS_GELU || epi == NBEST_EPI_BIAS_DROP_RES)
    NB_CHECK(a->bias, NBEST_ERR_ARG, "gemm: epilogue %d needs bias", epi);
  if (epi == NBEST_EPI_BIAS_DROP_RES || epi == NBEST_EPI_RES) NB_CHECK(a->R && a->ldr % 8 == 0 && ((uintptr_t)a->R & 15) == 0, NBEST_ERR_ARG, "gemm: epilogue %d needs R", epi);
  if (epi == NBEST_EPI_BIAS_GELU || epi == NBEST_EPI_DGELU) NB_CHECK(a->U && a->ldu % 8 == 0 && ((uintptr_t)a->U & 15) == 0, NBEST_ERR_ARG, "gemm: epilogue %d needs U", epi);
  if (epi == NBEST_EPI_F32_SPLITK && p.splits > 1)
    NB_CHECK(a->ws && a->ws_bytes >= (size_t)p.splits * a->M * a->N * sizeof(float), NBEST_ERR_WORKSPACE,
             "gemm: split-K workspace too small (%zu < %zu)", a->ws_bytes, (size_t)p.splits * a->M * a->N * sizeof(float));
  const int grid = p.tiles_m * p.tiles_n * p.splits;
  int rc;
  if (!a->trans_a && !a->trans_b) rc = launch_epi<false, false>(p, epi, grid, st);
  else if (!a->trans_a && a->trans_b) rc = launch_epi<false, true>(p, epi, grid, st);
  else rc = launch_epi<true, true>(p, epi, grid, st);
  if (rc) return rc;
  if (p.colpart) return nbest_internal_partial_rows_sum(p.colpart, p.tiles_m * 2, (int)a->N, a->colsum_out, a->colsum_accumulate, st);
  if (epi == NBEST_EPI_F32_SPLITK && p.splits > 1 && !(a->flags & NBEST_GEMM_DEFER_REDUCE)) {
    const int64_t MN = a->M * a->N;
    int64_t g = (MN / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    splitk_reduce_kernel<<<(int)g, 256, 0, st>>>(p.slab, (float*)a->C, MN, a->N, a->ldc, p.splits, a->accumulate);
    NB_LAUNCH_CHECK();
  }
  return NBEST_OK;
}
