// K2/K4/K6 in fp8: forward / dgrad GEMMs C = epi((A8 . W8^T) * out_scale) (gemm8_kernel) and weight gradients dW = dY8^T . X8
// (gemm8tt_kernel, second half of the file) with BOTH operands in OCP e4m3, on the block-scaled
// matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales; the per-tensor weight scale is applied to the
// fp32 accumulator in the epilogue).  On gfx950 the non-scaled fp8 MFMA runs at the bf16 rate; this one sustains
// 4.3 PFLOP/s in a register-only loop with random operands against 2.1 for v_mfma_f32_16x16x32_bf16
// (tools/micro/mx_probe.hip), i.e. twice the math per cycle at half the operand bytes per flop.
//
// Structure = the 256x256 ping-pong kernel of gemm_bf16_v2.hip, byte for byte: a stage is 64 k-BYTES (64 fp8 values
// instead of 32 bf16), so the LDS ring (4 stages x 32 KiB), the LDS-DMA issue count, the counted vmcnt waits and the
// two-group slot schedule are unchanged; a stage now feeds 8 MFMAs of K = 64 per wave instead of 32 of K = 32.
//   fragment: lane l holds row (l & 31) of a 32-row block and the 32 bytes [32 g, 32 g + 32) of its 64-byte k-row,
//             g = l >> 5 (any k order works as long as A and B use the same one - checked by tools/micro/mx_probe.py);
//   LDS image [rows][64 B]: 16-byte chunk index ^= (row >> 2) & 3 - conflict-free for ds_read_b128's four 16-lane groups
//             ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32): rows of one class r % 4 share 4 of the 16 slots
//             of the bank row, and within a group their r >> 2 are {0,3,5,6} or {1,2,4,7}: distinct in the low two bits;
//   C/D: with the operands swapped (B first) lane l owns output row (l & 31) and the columns
//        (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the 32-column block: four groups of 4 consecutive columns.
// Epilogues: forward NBEST_EPI_BIAS, NBEST_EPI_BIAS_GELU (gelu' 8-bit + the e4m3 copy of gelu the next GEMM reads; bf16 gelu only on
// request), NBEST_EPI_BIAS_DROP_RES; dgrad NBEST_EPI_NONE, NBEST_EPI_RES, NBEST_EPI_DGELU (x gelu', scaled e4m3 copy, amax, fused
// column sums).  Other outputs are bf16: attention, LayerNorm and the heads are the bf16 path.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr int BK8 = 64;
#ifndef NBEST_DIAG
#define NBEST_DIAG 0
#endif
// timing-only ablations of the epilogue (`make diag DIAG=<mask>`, results are wrong): 256 no GELU math, 512 no 8-bit stores,
// 1024 no bf16 store, 2048 no main loop; of the ping-pong main loop: 1 no in-loop DMA, 2 no fragment
// reads, 4 no MFMA
constexpr int DIAG8 = NBEST_DIAG;
// Epilogue stores are streaming (nontemporal, common.h st_stream): the outputs (50 - 400 MB per GEMM) otherwise wash the
// weights and the activation panel out of the 4 MiB L2 of every XCD while other tiles still read them (FFN-up 175 -> 163 us).

struct GemmP8 {
  const uint8_t* A; const uint8_t* B; bf16* C; const float* bias; const bf16* R; uint8_t* U; uint8_t* C8;
  int64_t M, N, K, lda, ldb, ldc, ldr, ldu, ldc8;
  int tiles_m, tiles_n;
  uint32_t a_bytes, b_bytes;
  float out_scale;
  const float* out_scale_dev;
  const uint32_t* a_amax;      // dgrad: A8 = e4m3(gradient * s), s from this amax (fp8_gscale_of): the accumulator is divided by s
  Fp8Grad c8g;                 // DGELU: e4m3 copy of the output gradient (scaled by ITS previous amax) + its new amax
  float* colpart;              // DGELU: fused column sums of the output (bias gradient), partial rows [tiles_m * 2][N]
  DropCfg drop;
  int stream_out;
  int gn;                      // tile columns per L2 group (common.h nb_tile_coords)
  const uint8_t* Bp;           // B pre-packed for this tile width (nbest_pack_weights_fp8): (tile column, K stage) images, contiguous; or nullptr
  uint32_t bp_bytes;
};

#ifdef NBEST_EXPERIMENTS
__device__ unsigned long long* g_trace8;   // per workgroup: key | late << 16, t_start, t_main_end, t_end (100 MHz clock)
#endif

__device__ __forceinline__ int xcd_remap8(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// one operand tile: ROWS rows x 64 bytes = 4 ROWS 16-byte chunks, NT threads -> 4 ROWS / NT LDS-DMA instructions per thread
template <int ROWS = 256, int NT = 512, int AUX = 0>
__device__ __forceinline__ void stage_tile8(__amdgpu_buffer_rsrc_t rs, char* tile, int64_t row0, int64_t k0, int64_t ld, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < ROWS * 4 / NT; ++i) {
    const int p = i * NT + tid;
    const int row = p >> 2, slot = p & 3;
    const int kc = slot ^ ((row >> 2) & 3);
    const uint32_t voff = (uint32_t)((row0 + row) * ld + k0 + kc * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (i * NT + wave * 64) * 16), 16, voff, 0, 0, AUX);
  }
}

// the same B tile from a PRE-PACKED operand (pack_b8_kernel): a linear copy, 1 KiB contiguous per wave-instruction
template <int ROWS, int NT, int AUX = 0>
__device__ __forceinline__ void stage_tile8_packed(__amdgpu_buffer_rsrc_t rs, char* tile, uint32_t stage_byte0, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < ROWS * 4 / NT; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (i * NT + wave * 64) * 16), 16, stage_byte0 + (uint32_t)(i * NT + tid) * 16u, 0, 0, AUX);
}

__device__ __forceinline__ i32x8 read_frag8(const char* tile, int row_base, int lane) {
  const int row = row_base + (lane & 31), g = lane >> 5, sw = (row >> 2) & 3;
  const i32x4 lo = *(const i32x4*)(tile + row * 64 + (((2 * g) ^ sw) << 4));
  const i32x4 hi = *(const i32x4*)(tile + row * 64 + (((2 * g + 1) ^ sw) << 4));
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int N> __device__ __forceinline__ void wait_vm8() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WN = 4: 256x256 tile, 8 waves, ONE workgroup per CU (4-stage ring = 128 KiB), the two wave groups ping-pong.
// WN = 2: 256x128 tile, 4 waves, TWO workgroups per CU (3-stage ring = 72 KiB each, <= 256 registers), no choreography
//         between them.  Pays 1.5x the LDS-DMA bytes per flop; ties with WN = 4 on the layer shapes except the small
//         N = K = 768 ones, where twice as many tiles fill the last round better.
template <int EPI, int WN>
__global__ __launch_bounds__(128 * WN, 2) void gemm8_kernel(GemmP8 p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int BM = 256, BN = 64 * WN, NT = 128 * WN, STAGES = (WN == 4) ? 4 : 3;
  constexpr int WTM = 128, WTN = 64, TMb = 4, TNb = 2;              // 32x32 blocks per wave tile
  constexpr int A_BYTES = BM * BK8, STAGE = A_BYTES + BN * BK8;     // 16 KiB + 16 (8) KiB
  constexpr int NDMA = (BM + BN) * 4 / NT;                          // LDS-DMA instructions per thread and stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int id = xcd_remap8(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  nb_tile_coords(id, p.tiles_m, p.gn, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
  const int nk = (DIAG8 & 2048) ? 1 : (int)(p.K / BK8);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const bool b_packed = p.Bp != nullptr;                        // workgroup-uniform
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(b_packed ? (void*)p.Bp : (void*)p.B, 0, b_packed ? p.bp_bytes : p.b_bytes, 0x00020000);
  const int nkt = (int)(p.K / BK8);
  constexpr int BNc = 64 * WN, B_BYTESc = BNc * BK8;

  constexpr bool kHasBias = (EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU || EPI == NBEST_EPI_BIAS_DROP_RES);
  constexpr bool kHasR = (EPI == NBEST_EPI_BIAS_DROP_RES || EPI == NBEST_EPI_RES);
  constexpr bool kHasUin = (EPI == NBEST_EPI_DGELU);
  const int64_t en8 = n0 + wn * WTN + (lane & 7) * 8;
  const int64_t erow0 = m0 + wm * WTM + (lane >> 3);
  f32x4 pb0 = {0, 0, 0, 0}, pb1 = {0, 0, 0, 0};
  if (kHasBias) { pb0 = *(const f32x4*)(p.bias + en8); pb1 = *(const f32x4*)(p.bias + en8 + 4); }
  float oscale = p.out_scale_dev ? *p.out_scale_dev : p.out_scale;
  // A8 = e4m3(a * s): a gradient (dgrad epilogues, scale target 56) or a forward activation (forward epilogues, target 224)
  constexpr bool kFwdEpi = (EPI == NBEST_EPI_BIAS || EPI == NBEST_EPI_BIAS_GELU || EPI == NBEST_EPI_BIAS_DROP_RES);
  if (p.a_amax) oscale /= kFwdEpi ? fp8_ascale_of(__uint_as_float(*p.a_amax)) : fp8_gscale_of(__uint_as_float(*p.a_amax));
  const float c8s = (EPI == NBEST_EPI_BIAS_GELU) ? fp8_act_scale(p.c8g.amax_prev) : fp8_grad_scale(p.c8g.amax_prev);
  float amax8 = 0.f;
  auto load_pre = [&](int64_t m) -> i32x4 {   // residual rows: bf16, 16 bytes per lane; GELU' rows: 8-bit, 8 bytes per lane
    if constexpr (kHasR) {
      return (m < p.M) ? *(const i32x4*)(p.R + m * p.ldr + en8) : i32x4{0, 0, 0, 0};
    } else {
      const i32x2 q = (m < p.M) ? *(const i32x2*)(p.U + m * p.ldu + en8) : i32x2{0, 0};
      return i32x4{q[0], q[1], 0, 0};
    }
  };
  i32x4 pre[TMb][4];
  if (kHasR || kHasUin) {
#pragma unroll
    for (int it = 0; it < 4; ++it) pre[0][it] = load_pre(erow0 + it * 8);
  }

  f32x16 acc[TMb][TNb];
#pragma unroll
  for (int i = 0; i < TMb; ++i)
#pragma unroll
    for (int j = 0; j < TNb; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  i32x8 af[TMb], bfr[TNb];
  if constexpr (WN == 4) {
    // ---- ping-pong main loop (see gemm_bf16_v2.hip: the two waves of every SIMD run one slot out of phase) ----
    const int grp = __builtin_amdgcn_readfirstlane(wm);
  #pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0) {
      if (s0 < nk) {
        stage_tile8<BM, NT, NB_AUX_A>(rsA, lds + s0 * STAGE, m0, (int64_t)s0 * BK8, p.lda, tid);
        if (b_packed) stage_tile8_packed<BN, NT, NB_AUX_B>(rsB, lds + s0 * STAGE + A_BYTES, (uint32_t)((tile_n * nkt + s0) * B_BYTESc), tid);
        else stage_tile8<BN, NT, NB_AUX_B>(rsB, lds + s0 * STAGE + A_BYTES, n0, (int64_t)s0 * BK8, p.ldb, tid);
      }
    }
    {
      const int younger = (nk - 1 < STAGES - 2) ? nk - 1 : STAGES - 2;
      if (younger >= 2) wait_vm8<2 * NDMA>();
      else if (younger == 1) wait_vm8<NDMA>();
      else wait_vm8<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // ---------------- LOAD slot ----------------
      if (!(DIAG8 & 1) && kt + STAGES - 1 < nk) {
        int nb = buf + STAGES - 1;
        if (nb >= STAGES) nb -= STAGES;
        const int64_t k0 = (int64_t)(kt + STAGES - 1) * BK8;
        stage_tile8<BM, NT, NB_AUX_A>(rsA, lds + nb * STAGE, m0, k0, p.lda, tid);
        if (b_packed) stage_tile8_packed<BN, NT, NB_AUX_B>(rsB, lds + nb * STAGE + A_BYTES, (uint32_t)((tile_n * nkt + kt + STAGES - 1) * B_BYTESc), tid);
        else stage_tile8<BN, NT, NB_AUX_B>(rsB, lds + nb * STAGE + A_BYTES, n0, k0, p.ldb, tid);
      }
      const char* cur = lds + buf * STAGE;
  #pragma unroll
      for (int j = 0; j < TNb; ++j) if (!(DIAG8 & 2) || kt == 0) bfr[j] = read_frag8(cur + A_BYTES, wn * WTN + j * 32, lane);
  #pragma unroll
      for (int i = 0; i < TMb; ++i) if (!(DIAG8 & 2) || kt == 0) af[i] = read_frag8(cur, wm * WTM + i * 32, lane);
      {
        const int c = (nk - 1 - kt < STAGES - 1) ? nk - 1 - kt : STAGES - 1;   // stages kt+1.. outstanding
        if (c >= 3) wait_vm8<2 * NDMA>();
        else if (c == 2) wait_vm8<NDMA>();
        else if (c == 1) wait_vm8<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA slot ----------------
      __builtin_amdgcn_s_setprio(1);
  #pragma unroll
      for (int i = 0; i < TMb; ++i)
  #pragma unroll
        for (int j = 0; j < TNb; ++j)
          if (!(DIAG8 & 4) || kt == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 127, 0, 127);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
    // ---- two independent workgroups per CU: plain 3-stage ring, one barrier per k-step ----
    // (The two workgroups of a CU drift out of phase by themselves - tools/gemm8_trace.py - so the epilogue of one does
    // overlap the main loop of the other; what it cannot buy back is that a main loop running alone has only its own two
    // stages in flight.)
#ifdef NBEST_EXPERIMENTS
    if (g_trace8 && tid == 0) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_trace8[blockIdx.x * 4 + 0] = ((unsigned long long)xcc << 32) | hw;
      g_trace8[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0) {
      if (s0 < nk) {
        stage_tile8<BM, NT, NB_AUX_A>(rsA, lds + s0 * STAGE, m0, (int64_t)s0 * BK8, p.lda, tid);
        if (b_packed) stage_tile8_packed<BN, NT, NB_AUX_B>(rsB, lds + s0 * STAGE + A_BYTES, (uint32_t)((tile_n * nkt + s0) * B_BYTESc), tid);
        else stage_tile8<BN, NT, NB_AUX_B>(rsB, lds + s0 * STAGE + A_BYTES, n0, (int64_t)s0 * BK8, p.ldb, tid);
      }
    }
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vm8<NDMA>();       // stage kt landed (stage kt + 1 may still be in flight)
      else wait_vm8<0>();
      __builtin_amdgcn_s_barrier();            // ... for every wave; and every wave is done reading stage kt - 1
      asm volatile("" ::: "memory");
      if (kt + STAGES - 1 < nk) {
        int nb = buf + STAGES - 1;
        if (nb >= STAGES) nb -= STAGES;
        const int64_t k0 = (int64_t)(kt + STAGES - 1) * BK8;
        stage_tile8<BM, NT, NB_AUX_A>(rsA, lds + nb * STAGE, m0, k0, p.lda, tid);
        if (b_packed) stage_tile8_packed<BN, NT, NB_AUX_B>(rsB, lds + nb * STAGE + A_BYTES, (uint32_t)((tile_n * nkt + kt + STAGES - 1) * B_BYTESc), tid);
        else stage_tile8<BN, NT, NB_AUX_B>(rsB, lds + nb * STAGE + A_BYTES, n0, k0, p.ldb, tid);
      }
      const char* cur = lds + buf * STAGE;
#pragma unroll
      for (int j = 0; j < TNb; ++j) bfr[j] = read_frag8(cur + A_BYTES, wn * WTN + j * 32, lane);
#pragma unroll
      for (int i = 0; i < TMb; ++i) af[i] = read_frag8(cur, wm * WTM + i * 32, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TMb; ++i)
#pragma unroll
        for (int j = 0; j < TNb; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 127, 0, 127);
      __builtin_amdgcn_s_setprio(0);
      buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    }
#ifdef NBEST_EXPERIMENTS
    if (g_trace8 && tid == 0) g_trace8[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
  }

  // ---- epilogue: 32-row blocks restaged through wave-private LDS ([32][64] fp32, chunk16 ^= row & 15) ----
  float* ep = (float*)lds + wave * 2048;
  float colacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (kHasR || kHasUin) {
#pragma unroll
    for (int c = 1; c < TMb; ++c)
#pragma unroll
      for (int it = 0; it < 4; ++it) pre[c][it] = load_pre(erow0 + c * 32 + it * 8);
  }
  __builtin_amdgcn_s_barrier();   // every wave has finished reading the operand ring
  asm volatile("" ::: "memory");
  const int hh = lane >> 5, wrow = lane & 31;
  // element offsets of this lane's 8-column group in the row-major outputs, advanced by 8 rows per iteration (computing
  // m * ld afresh for every row costs quarter-rate 64-bit multiplies: 15 % of the VALU work of the GELU epilogue)
  int64_t oC = erow0 * p.ldc + en8, oU = erow0 * p.ldu + en8, oC8 = erow0 * p.ldc8 + en8;
  const int64_t sC = 8 * p.ldc, sU = 8 * p.ldu, sC8 = 8 * p.ldc8;
  uint32_t dbase = (uint32_t)(erow0 * p.N + en8);
  const uint32_t sD = 8u * (uint32_t)p.N;
#pragma unroll
  for (int c = 0; c < TMb; ++c) {
#pragma unroll
    for (int j = 0; j < TNb; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cw = 8 * j + 2 * q + hh;   // 16-byte chunk: columns 32 j + 8 q + 4 hh .. + 3
        *(f32x4*)(ep + wrow * 64 + ((cw ^ (wrow & 15)) << 2)) =
            f32x4{acc[c][j][4 * q], acc[c][j][4 * q + 1], acc[c][j][4 * q + 2], acc[c][j][4 * q + 3]};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pidx = it * 64 + lane, row = pidx >> 3, c8 = pidx & 7;
      const int64_t m = m0 + wm * WTM + c * 32 + row;
      const f32x4 v0 = *(const f32x4*)(ep + row * 64 + (((2 * c8) ^ (row & 15)) << 2));
      const f32x4 v1 = *(const f32x4*)(ep + row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2));
      if (m < p.M) {
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = fmaf(v[e], oscale, pb0[e]); v[4 + e] = fmaf(v[4 + e], oscale, pb1[e]); }   // (no bias: pb = 0)
      if (EPI == NBEST_EPI_BIAS_GELU) {
        float gp[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          if (DIAG8 & 256) { gp[e] = v[e]; gp[e + 1] = v[e + 1]; continue; }
          f32x2 h2, g2;
          gelu_pair_fast(f32x2{v[e], v[e + 1]}, h2, g2);
          gp[e] = g2[0]; gp[e + 1] = g2[1]; v[e] = h2[0]; v[e + 1] = h2[1];
        }
        if (!(DIAG8 & 512)) {
          st_stream((i32x2*)(p.U + oU), i32x2{(int)gd_pack4(gp), (int)gd_pack4(gp + 4)}, p.stream_out);
          // e4m3 copy of gelu(u) for the FFN-down GEMM, times the tensor's delayed scale (c8s = 1 without a history); this pass's amax
          float q[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { q[e] = v[e] * c8s; amax8 = fmaxf(amax8, fabsf(v[e])); }
          st_stream((i32x2*)(p.C8 + oC8), i32x2{(int)fp8_pack4(q), (int)fp8_pack4(q + 4)}, p.stream_out);
        } else if (gp[0] + gp[3] + gp[5] == 123.f) p.U[0] = 1;
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
      if (kHasUin) {   // x GELU'(u); the result is the gradient the FFN-up dgrad / wgrad read: bf16 + (scaled) e4m3 copy + amax
        float gd[8];
        gd_unpack4((uint32_t)pre[c][it][0], gd);
        gd_unpack4((uint32_t)pre[c][it][1], gd + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] *= gd[e]; colacc[e] += v[e]; }
        if (p.c8g.amax_new) {
#pragma unroll
          for (int e = 0; e < 8; ++e) amax8 = fmaxf(amax8, fabsf(v[e]));
          if (p.c8g.out8) {
            float q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = v[e] * c8s;
            st_stream((i32x2*)(p.c8g.out8 + oC8), i32x2{(int)fp8_pack4(q), (int)fp8_pack4(q + 4)}, p.stream_out);
          }
        }
      }
      if ((EPI == NBEST_EPI_BIAS_GELU || EPI == NBEST_EPI_DGELU) && !p.C) {   // only the e4m3 copy is wanted
      } else if (!(DIAG8 & 1024)) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        st_stream((bf16x8*)(p.C + oC), o, p.stream_out);
      } else if (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] == 123.f) p.C[0] = (bf16)1.f;
      }
      oC += sC; oU += sU; oC8 += sC8; dbase += sD;
    }
    asm volatile("" ::: "memory");
  }
  if (kHasUin && p.colpart) {   // fused bias gradient: per-wave column sums -> partial rows (as gemm_bf16_v2.hip)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = colacc[e];
      x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
      colacc[e] = x;
    }
    if ((lane >> 3) == 0) {
      float* o = p.colpart + ((int64_t)tile_m * 2 + wm) * p.N + en8;
      *(f32x4*)o = f32x4{colacc[0], colacc[1], colacc[2], colacc[3]};
      *(f32x4*)(o + 4) = f32x4{colacc[4], colacc[5], colacc[6], colacc[7]};
    }
  }
  if ((kHasUin || EPI == NBEST_EPI_BIAS_GELU) && p.c8g.amax_new) {
    amax8 = wave_max(amax8);
    if (lane == 0) amax_update(p.c8g.amax_new, amax8);
  }
#ifdef NBEST_EXPERIMENTS
  if (WN == 2 && g_trace8 && tid == 0) g_trace8[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// bf16 [n] -> e4m3 (unit scale): the A operands of the fp8 forward GEMMs that no producer kernel writes directly
__global__ __launch_bounds__(256) void cast_bf16_fp8_kernel(const bf16* __restrict__ src, uint8_t* __restrict__ dst, int64_t n,
                                                            const uint32_t* __restrict__ a_prev, uint32_t* __restrict__ a_new) {
  const float s8 = fp8_act_scale(a_prev);        // delayed per-tensor activation scale (1 without a history); a_new: this pass's amax
  float amax8 = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 256 * 8) {
    float v[8];
    Vec8<bf16>::load(src + i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { amax8 = fmaxf(amax8, fabsf(v[e])); v[e] *= s8; }
    *(i32x2*)(dst + i) = i32x2{(int)fp8_pack4(v), (int)fp8_pack4(v + 4)};
  }
  if (a_new) {
    amax8 = wave_max(amax8);
    if ((threadIdx.x & 63) == 0) amax_update(a_new, amax8);
  }
}

// per-matrix quantisation of the weights: w8 = e4m3(w * scale), scale = 2^floor(log2(224 / max|w|)) (a power of two: exact)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ w, const nbest_matrix_desc* __restrict__ descs,
                                                     uint32_t* __restrict__ amax_bits) {
  __shared__ float sm[16];
  const nbest_matrix_desc d = descs[blockIdx.y];
  const int64_t n = (int64_t)d.rows * d.cols;
  float mx = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    const f32x4 x = *(const f32x4*)(w + d.offset + i);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
  }
  mx = wave_max(mx);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    atomicMax(amax_bits + blockIdx.y, __float_as_uint(mx));   // non-negative floats order like their bit patterns
  }
}
__global__ __launch_bounds__(256) void quant_w8_kernel(const float* __restrict__ w, uint8_t* __restrict__ w8,
                                                       const nbest_matrix_desc* __restrict__ descs,
                                                       const uint32_t* __restrict__ amax_bits, float* __restrict__ inv_scale) {
  const nbest_matrix_desc d = descs[blockIdx.y];
  const int64_t n = (int64_t)d.rows * d.cols;
  const float amax = __uint_as_float(amax_bits[blockIdx.y]);
  const float scale = fp8_scale_of(amax);
  if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[blockIdx.y] = 1.f / scale;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 2048) {
    float v[8];
    Vec8<float>::load(w + d.offset + i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= scale;
    *(i32x2*)(w8 + d.offset + i) = i32x2{(int)fp8_pack4(v), (int)fp8_pack4(v + 4)};
  }
}

// the same quantisation, written TRANSPOSED ([cols][rows] at the matrix' offset): the k-contiguous B operand of the dgrads.
// One launch: block -> (matrix, 64x64 tile) through descs[].tile_start, as transpose_multi_kernel.
// (w8 != nullptr: the transposed copy is made from the already quantised bytes - a quarter of the master's traffic)
__global__ __launch_bounds__(256) void quant_w8t_kernel(const float* __restrict__ w, const uint8_t* __restrict__ w8,
                                                        uint8_t* __restrict__ w8t,
                                                        const nbest_matrix_desc* __restrict__ descs, int n,
                                                        const uint32_t* __restrict__ amax_bits) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[64][80];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const nbest_matrix_desc d = descs[lo];
  const float scale = fp8_scale_of(__uint_as_float(amax_bits[lo]));
  const int t = blockIdx.x - d.tile_start, tc = (d.cols + 63) / 64;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = w + d.offset;
  uint8_t* o = w8t + d.offset;
  if (w8 && r0 + 64 <= d.rows && c0 + 64 <= d.cols && ((d.rows | d.cols | d.offset) & 15) == 0) {   // 16-byte path (every encoder matrix)
    transpose_tile64<uint8_t>(w8 + d.offset, o, d.rows, d.cols, r0, c0, &tile[0][0]);
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    float v[4] = {(r < d.rows && c < d.cols) ? src[(int64_t)r * d.cols + c] * scale : 0.f, 0.f, 0.f, 0.f};
    tile[ty + 4 * i][tx] = (uint8_t)(fp8_pack4(v) & 0xFFu);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < d.cols && r < d.rows) o[(int64_t)c * d.rows + r] = tile[tx][ty + 4 * i];
  }
}

// e4m3 B operand ([N][K] bytes, k-contiguous) -> the order gemm8_kernel stages it: per (tile column, 64-byte K stage) the bn x 64 LDS image
// (16-byte chunk slot ^ ((row >> 2) & 3), as stage_tile8), contiguous.  desc.pad = bn (256 | 128), tile_start = first block of the matrix.
__global__ __launch_bounds__(256) void pack_b8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      const nbest_matrix_desc* __restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const nbest_matrix_desc d = descs[lo];
  const int bn = d.pad, nk = d.cols / BK8, t = blockIdx.x - d.tile_start, tile_n = t / nk, kt = t - tile_n * nk;
  const uint8_t* s = src + d.offset + ((int64_t)tile_n * bn) * d.cols + (int64_t)kt * BK8;
  uint8_t* o = dst + d.offset + ((int64_t)tile_n * nk + kt) * bn * BK8;
  for (int p = threadIdx.x; p < bn * 4; p += blockDim.x) {
    const int row = p >> 2, slot = p & 3;
    const int kc = slot ^ ((row >> 2) & 3);
    *(i32x4*)(o + (int64_t)p * 16) = *(const i32x4*)(s + (int64_t)row * d.cols + kc * 16);
  }
}

// ---- weight gradient in fp8: dW[Nw][Kw] (fp32) = sum over tokens dY8[m][nw] * X8[m][kw] * out_scale ---------------------------------
// Both operands are TOKEN-major e4m3 tensors ([M][features], as their producers wrote them), the reduction runs over tokens, so the
// 32 k-bytes a lane feeds to the MFMA are a COLUMN of the LDS tile: ds_read_b64_tr_b8.  Its semantics (tools/micro/tr8_probe.hip):
// in a 16-lane group, lane i receives byte (i & 7) of the 8-byte chunks addressed by lanes 2j + (i >> 3), j = 0..7 - an 8x8 byte
// transpose per lane parity.  With source lane (j, p) of group G pointing at token row 32 (G >> 1) + 8 q + j, feature chunk
// 16 (G & 1) + 8 p, read q = 0..3 gives lane l tokens 32 (l >> 5) + 8 q .. + 7 of feature (l & 31): the MX operand.
// LDS image [64 tokens][256 B]: 16-byte chunk c of a token row sits in slot ((c & 1) << 3 | c >> 1) ^ (token & 7): the 8 rows
// of a transposed read hit 8 different slots, and the two 16-lane groups of a 32-lane half (chunks c, c + 1) opposite
// 128-byte halves of the bank row - conflict-free (with c ^ (token & 7) they met on the same half: 2-way).
// Same ping-pong schedule and 4-stage ring as gemm8_kernel (a stage = 64 tokens x 256 features per operand = 16 KiB);
// split-K over tokens into fp32 slabs, reduced by the caller (nbest_internal_splitk_reduce).
struct GemmP8T {
  const uint8_t* A; const uint8_t* B; float* C; float* slab;
  int64_t M, N, K, lda, ldb, ldc;           // M, N: output rows / cols (features of A / of B); K: tokens
  int64_t k_per_split;
  int tiles_m, tiles_n, splits, accumulate;
  uint32_t a_bytes, b_bytes;
  const uint32_t* a_amax;                    // dY8 = e4m3(dY * s): the accumulator is divided by s
  const uint32_t* b_amax;                    // X8 = e4m3(X * s_x) (forward activation copy, delayed scale): ... and by s_x; NULL = 1
  // second problem of a PAIR (nbest_wgrad_fp8_pair): output rows >= m_split of the virtual [M][N] result are A2^T . B2 with their own
  // gradient scale (same N and K); m_split = 0: none
  const uint8_t* A2; const uint8_t* B2;
  int64_t lda2, ldb2, m_split;
  uint32_t a2_bytes, b2_bytes;
  const uint32_t* a_amax2;
  const uint32_t* b_amax2;
};

__device__ __forceinline__ void stage_tile8t(__amdgpu_buffer_rsrc_t rs, char* tile, int64_t f0, int64_t k0, int64_t ld, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = i * 512 + tid;                 // 64 token rows x 16 chunks of 16 bytes
    const int row = p >> 4, slot = p & 15;
    const int t = slot ^ (row & 7);
    const int c = ((t & 7) << 1) | (t >> 3);     // slot = (chunk parity << 3 | chunk >> 1) ^ (token & 7)
    const uint32_t voff = (uint32_t)((k0 + row) * ld + f0 + c * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (i * 512 + wave * 64) * 16), 16, voff, 0, 0, 0);
  }
}

__device__ __forceinline__ i32x2 ds_read_tr8_asm(const char* addr) {
  i32x2 v;
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)addr;
  asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}

// features [f0, f0 + 32) x the 64 tokens of the stage -> MX operand (lane l: feature f0 + (l & 31), tokens 32 (l >> 5) .. + 31)
__device__ __forceinline__ i32x8 read_frag8t(const char* tile, int f0, int lane) {
  const int G = lane >> 4, i = lane & 15, j = i >> 1, pp = i & 1;
  const int chunk = (f0 >> 4) + (G & 1);
  i32x8 out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 32 * (G >> 1) + 8 * q + j;
    const i32x2 v = ds_read_tr8_asm(tile + row * 256 + (((((chunk & 1) << 3) | (chunk >> 1)) ^ (row & 7)) << 4) + 8 * pp);
    out[2 * q] = v[0]; out[2 * q + 1] = v[1];
  }
  return out;
}

// fragment reads of the stripped loop: one per-lane address register per 32-feature fragment + immediates (ring slot B_: + B_ KiB; read q:
// 8 token rows = two row groups further)
template <int OFF> __device__ __forceinline__ i32x2 ds_read_tr8_off(uint32_t a) {
  i32x2 v;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
template <int B_, int STAGES, int NF>
__device__ __forceinline__ void tt8_frags(const uint32_t* base, i32x8* fr) {
  constexpr int Q = 2 * STAGES * 1024;
#pragma unroll
  for (int t = 0; t < NF; ++t) {
    const i32x2 v0 = ds_read_tr8_off<B_ * 1024>(base[t]);
    const i32x2 v1 = ds_read_tr8_off<B_ * 1024 + Q>(base[t]);
    const i32x2 v2 = ds_read_tr8_off<B_ * 1024 + 2 * Q>(base[t]);
    const i32x2 v3 = ds_read_tr8_off<B_ * 1024 + 3 * Q>(base[t]);
    fr[t] = i32x8{v0[0], v0[1], v1[0], v1[1], v2[0], v2[1], v3[0], v3[1]};
  }
}

__global__ __launch_bounds__(512, 2) void gemm8tt_kernel(GemmP8T p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int BM = 256, BN = 256, WN = 4, STAGES = 4;
  constexpr int WTM = 128, WTN = 64, TMb = 4, TNb = 2;
  constexpr int A_BYTES = BK8 * BM, STAGE = 2 * A_BYTES;
  constexpr int NDMA = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int id = xcd_remap8(blockIdx.x, gridDim.x);
  const int tiles = p.tiles_m * p.tiles_n;
  const int z = id / tiles, t = id - z * tiles;
  const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
  const int64_t kbeg = (int64_t)z * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nk = (int)((kend - kbeg + BK8 - 1) / BK8);     // token rows past K are zero-filled by the buffer range check
  // operands of this tile (workgroup-uniform): the second problem's below m_split
  const bool second = p.m_split > 0 && m0 >= p.m_split;
  const int64_t m0a = second ? m0 - p.m_split : m0, lda_ = second ? p.lda2 : p.lda, ldb_ = second ? p.ldb2 : p.ldb;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(second ? (void*)p.A2 : (void*)p.A, 0, second ? p.a2_bytes : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(second ? (void*)p.B2 : (void*)p.B, 0, second ? p.b2_bytes : p.b_bytes, 0x00020000);
  const uint32_t* amaxp = second ? p.a_amax2 : p.a_amax;
  const uint32_t* bmaxp = second ? p.b_amax2 : p.b_amax;
  const float oscale = (amaxp ? 1.f / fp8_gscale_of(__uint_as_float(*amaxp)) : 1.f) * (bmaxp ? 1.f / fp8_ascale_of(__uint_as_float(*bmaxp)) : 1.f);

  f32x16 acc[TMb][TNb];
#pragma unroll
  for (int i = 0; i < TMb; ++i)
#pragma unroll
    for (int j = 0; j < TNb; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- main loop: the weight-gradient ping-pong of gemm_bf16_v2.hip (round 4), byte for byte in structure: LOAD slot stripped to its
  // memory instructions (wave index in an SGPR -> scalar LDS-DMA destinations; per-lane source offsets advanced by one v_add per piece
  // and slot - the token advance stays in the range-checked per-lane offset; 6 fragment-address registers set up once + immediates on a
  // ring image [16 groups of 4 token rows][stage][4 rows x 256 B] per operand), ONE workgroup barrier per stage (group 0: LOAD k | MFMA k,
  // group 1: MFMA k-1 | LOAD k), every counted wait before the barrier.  The generic loop it replaces re-derived the DMA offsets and the
  // 24 fragment addresses in every slot (stage_tile8t / read_frag8t on a runtime ring index).
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int grp = wv >> 2;             // = wm
  char* const ldsA = lds;
  char* const ldsB = lds + STAGES * A_BYTES;
  uint32_t voA[2], voB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = i * 512 + tid, row = pp >> 4, slot = pp & 15;
    const int t = slot ^ (row & 7);
    const int c = ((t & 7) << 1) | (t >> 3);
    voA[i] = (uint32_t)((kbeg + row) * lda_ + m0a + c * 16);
    voB[i] = (uint32_t)((kbeg + row) * ldb_ + n0 + c * 16);
  }
  const uint32_t stepA = (uint32_t)(BK8 * lda_), stepB = (uint32_t)(BK8 * ldb_);
  uint32_t faA[TMb], faB[TNb];
  {
    const int G = lane >> 4, i16 = lane & 15, j8 = i16 >> 1, pp = i16 & 1;
    const int row0 = 32 * (G >> 1) + j8;                          // token row of read q = 0 (q adds 8 rows = two row groups)
    const uint32_t rowpart = (uint32_t)((row0 >> 2) * (STAGES * 1024) + (row0 & 3) * 256 + 8 * pp);
    const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)ldsA;
    const uint32_t b0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)ldsB;
#pragma unroll
    for (int t = 0; t < TMb; ++t) {
      const int chunk = ((wm * WTM + t * 32) >> 4) + (G & 1);
      faA[t] = a0 + rowpart + (uint32_t)(((((chunk & 1) << 3) | (chunk >> 1)) ^ (row0 & 7)) << 4);
    }
#pragma unroll
    for (int t = 0; t < TNb; ++t) {
      const int chunk = ((wn * WTN + t * 32) >> 4) + (G & 1);
      faB[t] = b0 + rowpart + (uint32_t)(((((chunk & 1) << 3) | (chunk >> 1)) ^ (row0 & 7)) << 4);
    }
  }
  i32x8 af[TMb], bfr[TNb];
#define T8_ISSUE(NB_)                                                                                                                    \
  do {                                                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(ldsA + ((i * 8 + wv) * STAGES + (NB_)) * 1024), 16, voA[i], 0, 0, 0);        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(ldsB + ((i * 8 + wv) * STAGES + (NB_)) * 1024), 16, voB[i], 0, 0, 0);        \
    voA[0] += stepA; voA[1] += stepA; voB[0] += stepB; voB[1] += stepB;                                                                  \
  } while (0)
#define T8_FRAGS(B_) do { tt8_frags<(B_), STAGES, TNb>(faB, bfr); tt8_frags<(B_), STAGES, TMb>(faA, af); } while (0)
#define T8_READY()                                                                     \
  do {                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
    _Pragma("unroll") for (int i = 0; i < TMb; ++i) asm volatile("" : "+v"(af[i]));    \
    _Pragma("unroll") for (int j = 0; j < TNb; ++j) asm volatile("" : "+v"(bfr[j]));   \
  } while (0)
#define T8_LOAD_STEADY(B_) do { T8_ISSUE(((B_) + STAGES - 1) % STAGES); T8_FRAGS(B_); wait_vm8<(STAGES - 2) * NDMA>(); T8_READY(); } while (0)
#define T8_LOAD_GENERIC(KT, B_)                                                \
  do {                                                                         \
    const bool is_ = (KT) + STAGES - 1 < nk;                                   \
    if (is_) T8_ISSUE(((B_) + STAGES - 1) % STAGES);                           \
    T8_FRAGS(B_);                                                              \
    if ((KT) + 1 < nk) {                                                       \
      int full_ = nk - 2 - (KT);                                               \
      full_ = full_ < 0 ? 0 : (full_ > STAGES - 3 ? STAGES - 3 : full_);       \
      const int cnt_ = NDMA * full_ + (is_ ? NDMA : 0);                        \
      if (cnt_ >= 8) wait_vm8<8>(); else if (cnt_ == 4) wait_vm8<4>(); else wait_vm8<0>(); \
    }                                                                          \
    T8_READY();                                                                \
  } while (0)
#define T8_MFMA()                                                                                                          \
  do {                                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TMb; ++i)                                                                        \
      _Pragma("unroll") for (int j = 0; j < TNb; ++j)                                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 127, 0, 127);       \
    __builtin_amdgcn_s_setprio(0);                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
  } while (0)
#define T8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  static_assert(STAGES == 4 && NDMA == 4, "gemm8tt ping-pong: four ring slots, two LDS-DMA pieces per operand, wave and stage");
  if (0 < nk) T8_ISSUE(0);
  if (1 < nk) T8_ISSUE(1);
  if (2 < nk) T8_ISSUE(2);
  {
    const int younger = (nk - 1 < STAGES - 2) ? nk - 1 : STAGES - 2;
    if (younger >= 2) wait_vm8<2 * NDMA>();
    else if (younger == 1) wait_vm8<NDMA>();
    else wait_vm8<0>();
  }
  __builtin_amdgcn_s_barrier();                 // stage 0 landed for everyone
  asm volatile("" ::: "memory");
  const int n_steady = nk - (STAGES - 1);
  if (grp == 0) {
    int kt = 0;
    for (; kt + STAGES <= n_steady; kt += STAGES) {
      T8_LOAD_STEADY(0); T8_MFMA(); T8_BAR();
      T8_LOAD_STEADY(1); T8_MFMA(); T8_BAR();
      T8_LOAD_STEADY(2); T8_MFMA(); T8_BAR();
      T8_LOAD_STEADY(3); T8_MFMA(); T8_BAR();
    }
    for (; kt < nk; kt += STAGES) {
      T8_LOAD_GENERIC(kt, 0); T8_MFMA(); T8_BAR();
      if (kt + 1 < nk) { T8_LOAD_GENERIC(kt + 1, 1); T8_MFMA(); T8_BAR(); }
      if (kt + 2 < nk) { T8_LOAD_GENERIC(kt + 2, 2); T8_MFMA(); T8_BAR(); }
      if (kt + 3 < nk) { T8_LOAD_GENERIC(kt + 3, 3); T8_MFMA(); T8_BAR(); }
    }
    T8_BAR();
  } else {
    if (0 < nk) T8_LOAD_GENERIC(0, 0);
    T8_BAR();
    int kt = 1;
    for (; kt + STAGES <= n_steady; kt += STAGES) {
      T8_MFMA(); T8_LOAD_STEADY(1); T8_BAR();
      T8_MFMA(); T8_LOAD_STEADY(2); T8_BAR();
      T8_MFMA(); T8_LOAD_STEADY(3); T8_BAR();
      T8_MFMA(); T8_LOAD_STEADY(0); T8_BAR();
    }
    for (; kt < nk; kt += STAGES) {
      T8_MFMA(); T8_LOAD_GENERIC(kt, 1); T8_BAR();
      if (kt + 1 < nk) { T8_MFMA(); T8_LOAD_GENERIC(kt + 1, 2); T8_BAR(); }
      if (kt + 2 < nk) { T8_MFMA(); T8_LOAD_GENERIC(kt + 2, 3); T8_BAR(); }
      if (kt + 3 < nk) { T8_MFMA(); T8_LOAD_GENERIC(kt + 3, 0); T8_BAR(); }
    }
    if (0 < nk) T8_MFMA();
    T8_BAR();
  }
#undef T8_ISSUE
#undef T8_FRAGS
#undef T8_READY
#undef T8_LOAD_STEADY
#undef T8_LOAD_GENERIC
#undef T8_MFMA
#undef T8_BAR

  float* ep = (float*)lds + wave * 2048;
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const int hh = lane >> 5, wrow = lane & 31;
  const int64_t en8 = n0 + wn * WTN + (lane & 7) * 8;
#pragma unroll
  for (int c = 0; c < TMb; ++c) {
#pragma unroll
    for (int j = 0; j < TNb; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cw = 8 * j + 2 * q + hh;
        *(f32x4*)(ep + wrow * 64 + ((cw ^ (wrow & 15)) << 2)) =
            f32x4{acc[c][j][4 * q], acc[c][j][4 * q + 1], acc[c][j][4 * q + 2], acc[c][j][4 * q + 3]};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pidx = it * 64 + lane, row = pidx >> 3, c8 = pidx & 7;
      const int64_t m = m0 + wm * WTM + c * 32 + row;
      f32x4 v0 = *(const f32x4*)(ep + row * 64 + (((2 * c8) ^ (row & 15)) << 2));
      f32x4 v1 = *(const f32x4*)(ep + row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2));
      if (m >= p.M) continue;
      v0 *= oscale; v1 *= oscale;
      float* cp = (p.splits > 1) ? p.slab + ((int64_t)z * p.M + m) * p.N + en8 : p.C + m * p.ldc + en8;
      if (p.splits == 1 && p.accumulate) { v0 += *(const f32x4*)cp; v1 += *(const f32x4*)(cp + 4); }
      *(f32x4*)cp = v0;
      *(f32x4*)(cp + 4) = v1;
    }
    asm volatile("" ::: "memory");
  }
}

__global__ __launch_bounds__(256) void splitk_reduce8_kernel(const float* __restrict__ slab, float* __restrict__ C, int64_t MN,
                                                             int64_t N, int64_t ldc, int splits, int accumulate,
                                                             float* __restrict__ C2, int64_t m_split, int64_t ldc2) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < MN; i += (int64_t)gridDim.x * blockDim.x * 4) {
    f32x4 s = *(const f32x4*)(slab + i);
    for (int z = 1; z < splits; ++z) s += *(const f32x4*)(slab + (int64_t)z * MN + i);
    const int64_t m = i / N, n = i - m * N;
    float* c = (m < m_split) ? C + m * ldc + n : C2 + (m - m_split) * ldc2 + n;   // rows >= m_split: the second output of a pair
    if (accumulate) s += *(const f32x4*)c;
    *(f32x4*)c = s;
  }
}

static void plan8tt(int64_t M, int64_t N, int64_t K, int* splits, int64_t* kps) {
  const int64_t tiles = (M / 256) * (N / 256);
  const int64_t maxs = (K / 512 < 1) ? 1 : ((K / 512 > 32) ? 32 : K / 512);
  int64_t best_s = 1;
  double best = -1.0;
  for (int64_t sp = 1; sp <= maxs; ++sp) {
    const int64_t blocks = tiles * sp;
    const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
    if (eff > best + 1e-9) { best = eff; best_s = sp; }
    if (blocks * 5 >= 256 * 4 && eff >= 0.93) { best_s = sp; break; }
  }
  int64_t k = (K + best_s - 1) / best_s;
  k = (k + 63) / 64 * 64;
  *splits = (int)((K + k - 1) / k);
  *kps = k;
}

__global__ __launch_bounds__(256) void amax_bf16_kernel(const bf16* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
  float mx = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 2048) {
    float v[8];
    Vec8<bf16>::load(x + i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[e]));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) amax_update(out, mx);
}

}  // namespace

// *out = max(*out, max |x|) over a bf16 tensor (calibration pass of the fp8 dgrads: tensors whose producer is a bf16 kernel)
int nbest_internal_amax_bf16(const void* x, int64_t n, uint32_t* out, hipStream_t st) {
  int64_t g = (n / 8 + 255) / 256;
  if (g > 2048) g = 2048;
  amax_bf16_kernel<<<(int)g, 256, 0, st>>>((const bf16*)x, n, out);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

namespace {
// out[t] = max over the slots of tensor t; the slots are zeroed for the next pass.  One wave per tensor.
__global__ __launch_bounds__(64) void amax_fold_kernel(uint32_t* __restrict__ slots, uint32_t* __restrict__ out, int n) {
  const int t = blockIdx.x, lane = threadIdx.x;
  if (t >= n) return;
  uint32_t* s = slots + (int64_t)t * NBEST_AMAX_TENSOR_WORDS + (lane & (kAmaxSlots - 1)) * kAmaxSlotStride;
  float v = 0.f;
  if (lane < kAmaxSlots) { v = __uint_as_float(*s); *s = 0u; }        // (float bits of values >= 0 order like the floats)
  v = wave_max(v);
  if (lane == 0) out[t] = __float_as_uint(v);
}
}  // namespace

extern "C" int nbest_fp8_amax_fold(void* slots, void* out, int32_t n_tensors, void* stream) {
  NB_CHECK(slots && out && n_tensors >= 0, NBEST_ERR_ARG, "nbest_fp8_amax_fold: null argument");
  if (n_tensors == 0) return NBEST_OK;
  amax_fold_kernel<<<n_tensors, 64, 0, (hipStream_t)stream>>>((uint32_t*)slots, (uint32_t*)out, n_tensors);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" size_t nbest_wgrad_fp8_ws_bytes(int64_t M, int64_t N, int64_t K) {
  int sp; int64_t kps;
  plan8tt(M, N, K, &sp, &kps);
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}

// dW[M][N] (fp32, ldc) (+)= sum_k dY8[k][M-features] * X8[k][N-features] / s(a_amax); dY8 [K tokens][lda], X8 [K tokens][ldb] e4m3
// second problem (pair): dY8b / X8b / dWb / Mb rows / its own scale, appended below the first's output tiles
static int wgrad_fp8_impl(const void* dY8, const void* X8, float* dW, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                          int64_t ldc, const uint32_t* a_amax, const uint32_t* x_amax, const void* dY8b, const void* X8b, float* dWb, int64_t Mb,
                          int64_t ldab, int64_t ldbb, int64_t ldcb, const uint32_t* a_amaxb, const uint32_t* x_amaxb, int accumulate, void* ws,
                          size_t ws_bytes, hipStream_t st) {
  const int64_t Mv = M + Mb;     // rows of the virtual output
  GemmP8T p;
  p.A = (const uint8_t*)dY8; p.B = (const uint8_t*)X8; p.C = dW; p.slab = (float*)ws;
  p.M = Mv; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  plan8tt(Mv, N, K, &p.splits, &p.k_per_split);
  p.tiles_m = (int)(Mv / 256); p.tiles_n = (int)(N / 256);
  p.accumulate = accumulate;
  const int64_t ab = (K - 1) * lda + M, bb = (K - 1) * ldb + N;
  NB_CHECK(ab < ((int64_t)1 << 32) && bb < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "wgrad_fp8: operand larger than 4 GiB");
  p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  p.a_amax = a_amax; p.b_amax = x_amax;
  p.A2 = p.B2 = nullptr; p.lda2 = p.ldb2 = p.m_split = 0; p.a2_bytes = p.b2_bytes = 0; p.a_amax2 = nullptr; p.b_amax2 = nullptr;
  if (Mb > 0) {
    const int64_t ab2 = (K - 1) * ldab + Mb, bb2 = (K - 1) * ldbb + N;
    NB_CHECK(ab2 < ((int64_t)1 << 32) && bb2 < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "wgrad_fp8: operand larger than 4 GiB");
    NB_CHECK(p.splits > 1, NBEST_ERR_SHAPE, "wgrad_fp8_pair: needs a split-K plan");
    p.A2 = (const uint8_t*)dY8b; p.B2 = (const uint8_t*)X8b; p.lda2 = ldab; p.ldb2 = ldbb; p.m_split = M;
    p.a2_bytes = (uint32_t)ab2; p.b2_bytes = (uint32_t)bb2; p.a_amax2 = a_amaxb; p.b_amax2 = x_amaxb;
  }
  if (p.splits > 1) NB_CHECK(ws && ws_bytes >= (size_t)p.splits * Mv * N * sizeof(float), NBEST_ERR_WORKSPACE, "wgrad_fp8: workspace too small");
  constexpr int lds_bytes = 4 * 2 * 256 * BK8;
  (void)hipFuncSetAttribute((const void*)gemm8tt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  gemm8tt_kernel<<<p.tiles_m * p.tiles_n * p.splits, 512, lds_bytes, st>>>(p);
  NB_LAUNCH_CHECK();
  if (p.splits > 1) {
    const int64_t MN = Mv * N;
    int64_t g = (MN / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    splitk_reduce8_kernel<<<(int)g, 256, 0, st>>>(p.slab, dW, MN, N, ldc, p.splits, accumulate, dWb, Mb > 0 ? M : Mv, ldcb);
    NB_LAUNCH_CHECK();
  }
  return NBEST_OK;
}

extern "C" int nbest_wgrad_fp8(const void* dY8, const void* X8, float* dW, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                               int64_t ldc, const uint32_t* a_amax, const uint32_t* x_amax, int accumulate, void* ws, size_t ws_bytes,
                               nbest_stream_t stream) {
  NB_CHECK(dY8 && X8 && dW && M > 0 && N > 0 && K > 0, NBEST_ERR_ARG, "wgrad_fp8: bad arguments");
  NB_CHECK(M % 256 == 0 && N % 256 == 0, NBEST_ERR_SHAPE, "wgrad_fp8: output %lld x %lld must be multiples of 256", (long long)M, (long long)N);
  NB_CHECK(lda % 16 == 0 && ldb % 16 == 0 && ldc % 8 == 0 && ((uintptr_t)dY8 & 15) == 0 && ((uintptr_t)X8 & 15) == 0 && ((uintptr_t)dW & 15) == 0,
           NBEST_ERR_ALIGN, "wgrad_fp8: alignment");
  return wgrad_fp8_impl(dY8, X8, dW, M, N, K, lda, ldb, ldc, a_amax, x_amax, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, accumulate, ws,
                        ws_bytes, (hipStream_t)stream);
}

// two fp8 weight gradients with the same K (tokens) and N in one launch (the e4m3 counterpart of nbest_wgrad_pair): the 256 x 256 output
// tiles of the second problem are appended below the first's, each problem keeps its own gradient scale; one reduce writes both outputs
extern "C" size_t nbest_wgrad_fp8_pair_ws_bytes(int64_t Ma, int64_t Mb, int64_t N, int64_t K) {
  if (Ma <= 0 || Mb <= 0 || Ma % 256 || Mb % 256 || N % 256) return 0;
  int sp; int64_t kps;
  plan8tt(Ma + Mb, N, K, &sp, &kps);
  return sp > 1 ? (size_t)sp * (Ma + Mb) * N * sizeof(float) : 0;
}
extern "C" int nbest_wgrad_fp8_pair(const void* dY8a, const void* X8a, float* dWa, int64_t Ma, int64_t lda_a, int64_t ldb_a, int64_t ldc_a,
                                    const uint32_t* amax_a, const uint32_t* xamax_a, const void* dY8b, const void* X8b, float* dWb, int64_t Mb,
                                    int64_t lda_b, int64_t ldb_b, int64_t ldc_b, const uint32_t* amax_b, const uint32_t* xamax_b, int64_t N,
                                    int64_t K, int accumulate, void* ws, size_t ws_bytes, nbest_stream_t stream) {
  NB_CHECK(dY8a && X8a && dWa && dY8b && X8b && dWb && N > 0 && K > 0, NBEST_ERR_ARG, "wgrad_fp8_pair: bad arguments");
  NB_CHECK(nbest_wgrad_fp8_pair_ws_bytes(Ma, Mb, N, K) > 0, NBEST_ERR_SHAPE, "wgrad_fp8_pair: Ma, Mb, N must be multiples of 256 and the pair must split K");
  NB_CHECK(lda_a % 16 == 0 && ldb_a % 16 == 0 && ldc_a % 8 == 0 && lda_b % 16 == 0 && ldb_b % 16 == 0 && ldc_b % 8 == 0 &&
               (((uintptr_t)dY8a | (uintptr_t)X8a | (uintptr_t)dWa | (uintptr_t)dY8b | (uintptr_t)X8b | (uintptr_t)dWb) & 15) == 0,
           NBEST_ERR_ALIGN, "wgrad_fp8_pair: alignment");
  return wgrad_fp8_impl(dY8a, X8a, dWa, Ma, N, K, lda_a, ldb_a, ldc_a, amax_a, xamax_a, dY8b, X8b, dWb, Mb, lda_b, ldb_b, ldc_b, amax_b, xamax_b,
                        accumulate, ws, ws_bytes, (hipStream_t)stream);
}

// tile width gemm8_kernel uses for an [N][K] e4m3 weight at training-size token counts (its wn rule at M = 32 768)
extern "C" int nbest_pack_bn_fp8(int64_t N, int64_t K) {
  if (N % 256) return 0;
  return (N <= 768 && K <= 1024) ? 128 : 256;
}
extern "C" int nbest_pack_weights_fp8(const void* src, void* dst, const nbest_matrix_desc* descs, int n_matrices, int n_stages,
                                      nbest_stream_t stream) {
  NB_CHECK(src && dst && descs && n_matrices > 0 && n_stages > 0 && src != dst, NBEST_ERR_ARG, "pack_weights_fp8: bad arguments");
  pack_b8_kernel<<<n_stages, 256, 0, (hipStream_t)stream>>>((const uint8_t*)src, (uint8_t*)dst, descs, n_matrices);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

int nbest_internal_cast_bf16_to_fp8(const void* src, void* dst, int64_t n, const uint32_t* a_prev, uint32_t* a_new, hipStream_t st) {
  NB_CHECK(src && dst && n > 0 && n % 8 == 0, NBEST_ERR_ARG, "cast_bf16_to_fp8: bad arguments");
  int64_t g = (n / 8 + 255) / 256;
  if (g > 4096) g = 4096;
  cast_bf16_fp8_kernel<<<(int)g, 256, 0, st>>>((const bf16*)src, (uint8_t*)dst, n, a_prev, a_new);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}
extern "C" int nbest_cast_bf16_to_fp8(const void* src, void* dst, int64_t n, nbest_stream_t stream) {
  return nbest_internal_cast_bf16_to_fp8(src, dst, n, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int nbest_quantize_weights_fp8(const float* master, void* w8, void* w8t, const nbest_matrix_desc* descs, int n_matrices,
                                          int n_tiles, float* inv_scale, void* ws, size_t ws_bytes, nbest_stream_t stream) {
  NB_CHECK(master && w8 && descs && inv_scale && ws && n_matrices > 0, NBEST_ERR_ARG, "quantize_weights_fp8: bad arguments");
  NB_CHECK(ws_bytes >= (size_t)n_matrices * sizeof(uint32_t), NBEST_ERR_WORKSPACE, "quantize_weights_fp8: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  NB_CHECK(hipMemsetAsync(ws, 0, (size_t)n_matrices * sizeof(uint32_t), st) == hipSuccess, NBEST_ERR_LAUNCH, "quantize_weights_fp8: memset failed");
  absmax_kernel<<<dim3(64, n_matrices), 256, 0, st>>>(master, descs, (uint32_t*)ws);
  NB_LAUNCH_CHECK();
  quant_w8_kernel<<<dim3(64, n_matrices), 256, 0, st>>>(master, (uint8_t*)w8, descs, (const uint32_t*)ws, inv_scale);
  NB_LAUNCH_CHECK();
  if (w8t) {
    NB_CHECK(n_tiles > 0, NBEST_ERR_ARG, "quantize_weights_fp8: n_tiles");
    quant_w8t_kernel<<<n_tiles, 256, 0, st>>>(master, (const uint8_t*)w8, (uint8_t*)w8t, descs, n_matrices, (const uint32_t*)ws);
    NB_LAUNCH_CHECK();
  }
  return NBEST_OK;
}

int nbest_internal_partial_rows_sum(const float* part, int nrows, int N, float* out, int accumulate, hipStream_t st);

#ifdef NBEST_EXPERIMENTS
extern "C" int nbest_experiment_trace8(void* buf) {   // nullptr: off
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace8), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t nbest_gemm_fp8_ws_bytes(const nbest_gemm_fp8_args* a) {
  return (a && a->colsum_out) ? (size_t)((a->M + 255) / 256) * 2 * a->N * sizeof(float) : 0;
}

extern "C" int nbest_gemm_fp8(const nbest_gemm_fp8_args* a, nbest_stream_t stream) {
  NB_CHECK(a && a->A && a->B, NBEST_ERR_ARG, "gemm_fp8: null pointer");
  // the bf16 output may be dropped where the epilogue also writes the e4m3 copy its only readers take
  NB_CHECK(a->C || ((a->epilogue == NBEST_EPI_BIAS_GELU || a->epilogue == NBEST_EPI_DGELU) && a->C8), NBEST_ERR_ARG, "gemm_fp8: null output");
  NB_CHECK(a->M > 0 && a->N % 256 == 0 && a->K % BK8 == 0 && a->K >= BK8, NBEST_ERR_SHAPE,
           "gemm_fp8: needs N %% 256 == 0 and K %% 64 == 0 (M=%lld N=%lld K=%lld)", (long long)a->M, (long long)a->N, (long long)a->K);
  NB_CHECK(a->lda % 16 == 0 && a->ldb % 16 == 0 && a->ldc % 8 == 0, NBEST_ERR_ALIGN, "gemm_fp8: leading dimensions");
  NB_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0 && ((uintptr_t)a->C & 15) == 0, NBEST_ERR_ALIGN,
           "gemm_fp8: pointers must be 16-byte aligned");
  const int epi = a->epilogue;
  NB_CHECK(epi == NBEST_EPI_NONE || epi == NBEST_EPI_BIAS || epi == NBEST_EPI_BIAS_GELU || epi == NBEST_EPI_BIAS_DROP_RES ||
               epi == NBEST_EPI_DGELU || epi == NBEST_EPI_RES, NBEST_ERR_ARG, "gemm_fp8: epilogue %d is not built", epi);
  if (epi == NBEST_EPI_BIAS || epi == NBEST_EPI_BIAS_GELU || epi == NBEST_EPI_BIAS_DROP_RES)
    NB_CHECK(a->bias, NBEST_ERR_ARG, "gemm_fp8: epilogue %d needs bias", epi);
  if (epi == NBEST_EPI_BIAS_GELU)
    NB_CHECK(a->U && a->C8 && a->ldu % 8 == 0 && a->ldc8 % 8 == 0 && ((uintptr_t)a->U & 7) == 0 && ((uintptr_t)a->C8 & 7) == 0,
             NBEST_ERR_ARG, "gemm_fp8: BIAS_GELU needs U (8-bit gelu') and C8 (fp8 copy of the output)");
  if (epi == NBEST_EPI_BIAS_DROP_RES || epi == NBEST_EPI_RES)
    NB_CHECK(a->R && a->ldr % 8 == 0 && ((uintptr_t)a->R & 15) == 0, NBEST_ERR_ARG, "gemm_fp8: epilogue %d needs R", epi);
  if (epi == NBEST_EPI_DGELU) {
    NB_CHECK(a->U && a->ldu % 8 == 0 && ((uintptr_t)a->U & 7) == 0, NBEST_ERR_ARG, "gemm_fp8: DGELU needs U (8-bit gelu')");
    NB_CHECK(!a->C8 || (a->ldc8 % 8 == 0 && ((uintptr_t)a->C8 & 7) == 0), NBEST_ERR_ARG, "gemm_fp8: DGELU fp8 output alignment");
    NB_CHECK(!a->colsum_out || (a->ws && a->ws_bytes >= nbest_gemm_fp8_ws_bytes(a)), NBEST_ERR_WORKSPACE, "gemm_fp8: column-sum workspace too small");
  }
  GemmP8 p;
  p.A = (const uint8_t*)a->A; p.B = (const uint8_t*)a->B; p.C = (bf16*)a->C; p.bias = a->bias; p.R = (const bf16*)a->R;
  p.U = (uint8_t*)a->U; p.C8 = (uint8_t*)a->C8;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldu = a->ldu; p.ldc8 = a->ldc8;
  // 256x128 tiles, two workgroups per CU, where 256x256 tiles leave the last round half empty and the main loop is short
  // (N = K = 768: 384 big tiles on 256 CUs; 47 -> 41 us); the big ping-pong tile elsewhere (the two tie on the other shapes)
  int wn = (a->N * (int64_t)((a->M + 255) / 256) <= 3 * 128 * 256 && a->K <= 1024) ? 2 : 4;
#ifdef NBEST_EXPERIMENTS
  if (const char* e = getenv("NBEST_GEMM8_WN")) if (e[0] == '2' || e[0] == '4') wn = e[0] - '0';
#endif
  p.tiles_m = (int)((a->M + 255) / 256);
  p.tiles_n = (int)(a->N / (64 * wn));
  p.Bp = nullptr; p.bp_bytes = 0;
  if (a->B_packed && a->b_pack_bn == 64 * wn && a->ldb == a->K && a->N * a->K < ((int64_t)1 << 32) && ((uintptr_t)a->B_packed & 15) == 0) {
    p.Bp = (const uint8_t*)a->B_packed;
    p.bp_bytes = (uint32_t)(a->N * a->K);
  }
  const int64_t ab = (a->M - 1) * a->lda + a->K, bb = (a->N - 1) * a->ldb + a->K;
  NB_CHECK(ab < ((int64_t)1 << 32) && bb < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "gemm_fp8: operand larger than 4 GiB");
  p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  p.out_scale = a->out_scale;
  p.out_scale_dev = a->out_scale_dev;
  p.a_amax = a->a_amax;
  // DGELU: C8 = scaled e4m3 copy of the gradient; BIAS_GELU: p.C8 = e4m3 copy of gelu(u), scaled the same way when a history is given
  p.c8g = Fp8Grad{epi == NBEST_EPI_DGELU ? (uint8_t*)a->C8 : nullptr, a->c8_amax_prev, a->c8_amax_new};
  p.colpart = (epi == NBEST_EPI_DGELU && a->colsum_out) ? (float*)a->ws : nullptr;
  p.drop = make_drop(a->drop_p, a->seed, a->drop_stream);
  p.stream_out = nb_stream_output(a->M * a->N * 2) ? 1 : 0;
  p.gn = nb_group_cols(p.tiles_n, (int64_t)64 * wn * a->K, 1600);
  NB_CHECK(a->M * a->N < ((int64_t)1 << 32) || p.drop.thr16 == 0, NBEST_ERR_SHAPE, "gemm_fp8: dropout counter overflow");

  const int grid = p.tiles_m * p.tiles_n;
  constexpr int lds4 = 4 * (256 + 256) * BK8, lds2 = 3 * (256 + 128) * BK8;
  hipStream_t st = (hipStream_t)stream;
#define L8(E)                                                                                                          \
  case E:                                                                                                              \
    if (wn == 4) {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)gemm8_kernel<E, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);    \
      gemm8_kernel<E, 4><<<grid, 512, lds4, st>>>(p);                                                                  \
    } else {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)gemm8_kernel<E, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);    \
      gemm8_kernel<E, 2><<<grid, 256, lds2, st>>>(p);                                                                  \
    }                                                                                                                  \
    break;
  switch (epi) {
    L8(NBEST_EPI_NONE) L8(NBEST_EPI_BIAS) L8(NBEST_EPI_BIAS_GELU) L8(NBEST_EPI_BIAS_DROP_RES) L8(NBEST_EPI_DGELU) L8(NBEST_EPI_RES)
    default: break;
  }
#undef L8
  NB_LAUNCH_CHECK();
  if (p.colpart) return nbest_internal_partial_rows_sum(p.colpart, p.tiles_m * 2, (int)a->N, a->colsum_out, a->colsum_accumulate, st);
  return NBEST_OK;
}
