// K3 / K3b: scaled-dot-product attention with a per-sample key-padding mask, forward and backward.
//
// bf16 path (production): one workgroup (4 waves) per (sample, head); the whole K and V of a head
// live in LDS (S <= 256, d = 64: 2 x 32 KiB; 256 < S <= 512: see the long-sequence kernels), filled by LDS-DMA (buffer_load ... lds) straight from the
// fused qkv activation [M][3H] - the sequence's own buffer descriptor zero-fills rows past S.
//   forward : wave = 32 query rows.  scores are computed SWAPPED, S^T = K . Q^T with
//             v_mfma_f32_32x32x16_bf16, so a lane owns ONE query row (half of its keys; the other
//             half sits in lane^32): softmax max/sum are in-lane reductions + one wave shuffle.
//             The probability tile never leaves registers: the fp32 accumulator tile, converted to
//             bf16, IS the A operand of the P.V MFMA (k order permuted; V is fetched with the
//             matching permuted transposed LDS read ds_read_b64_tr_b16).
//   backward: wave = 32 keys ("key on the lane"): S = Q.K^T and dP = dO.V^T tiles are recomputed per
//             32-query block from the saved log-sum-exp; dV += P^T dO and dK += dS^T Q use the
//             accumulator-as-operand trick again (no LDS round trip, no atomics, no cross-wave sums);
//             only dS crosses LDS once, for dQ = dS . K.
//   LDS image of every [rows][64] bf16 tile: 128-byte rows, 16-byte chunk index XORed with
//             g(row) = ((row>>2)&3) | (((row>>1)&1)<<2): conflict-free for the ds_read_b128 row reads
//             of the 32x32x16 operand AND for the 4-row transposed reads.
// fp32 path (parity/debug): one thread per query row, plain VALU math, any S <= 512.
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ int crow(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
__device__ __forceinline__ int gsw(int row) { return ((row >> 2) & 3) | (((row >> 1) & 1) << 2); }

// =================================================================================================
// fp32 reference-grade kernels
// =================================================================================================
__global__ __launch_bounds__(128) void attn_fwd_f32_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                           float* __restrict__ ctx, float* __restrict__ lse, int S, int heads,
                                                           int H, float scale, DropCfg drop) {
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int i = blockIdx.y * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const int64_t ld = 3 * (int64_t)H;
  const float* base = qkv + (int64_t)b * S * ld;
  const float* qp = base + i * ld + h * 64;
  float q[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) q[d] = qp[d];
  float mx = -INFINITY, l = 0.f;
  for (int j = 0; j < S; ++j) {
    if (!mask[b * S + j]) continue;
    const float* kp = base + j * ld + H + h * 64;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) s = fmaf(q[d], kp[d], s);
    s *= scale;
    const float nm = fmaxf(mx, s);
    l = l * expf(mx - nm) + expf(s - nm);
    mx = nm;
  }
  const float lse_i = mx + logf(l);
  float o[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.f;
  for (int j = 0; j < S; ++j) {
    if (!mask[b * S + j]) continue;
    const float* kp = base + j * ld + H + h * 64;
    const float* vp = base + j * ld + 2 * H + h * 64;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) s = fmaf(q[d], kp[d], s);
    float p = expf(s * scale - lse_i);
    if (drop.thr16) p = nb_keep(drop, (uint32_t)((bh * S + i) * S + j)) ? p * drop.scale : 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = fmaf(p, vp[d], o[d]);
  }
  float* op = ctx + ((int64_t)b * S + i) * H + h * 64;
#pragma unroll
  for (int d = 0; d < 64; ++d) op[d] = o[d];
  lse[(int64_t)bh * S + i] = lse_i;
}

// fp32 parity path, query on the thread: dQ only.  (Until round 4 this kernel also added its dK / dV contributions with float atomics -
// the one place left where the fp32 step was not reproducible run to run: the 6-epoch F1 trajectories of tests/test_text_pipeline.py
// moved by +-1 pt between two runs of the same code.  dK / dV: attn_bwd_f32_kv_kernel, key on the thread, fixed summation order.)
__global__ __launch_bounds__(128) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                           const float* __restrict__ lse, float* __restrict__ dqkv, int S,
                                                           int heads, int H, float scale, DropCfg drop) {
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int i = blockIdx.y * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const int64_t ld = 3 * (int64_t)H;
  const float* base = qkv + (int64_t)b * S * ld;
  float* dbase = dqkv + (int64_t)b * S * ld;
  const float* qp = base + i * ld + h * 64;
  const float* dop = dctx + ((int64_t)b * S + i) * H + h * 64;
  const float* op = ctx + ((int64_t)b * S + i) * H + h * 64;
  float q[64], dO[64], dq[64];
  float delta = 0.f;
#pragma unroll
  for (int d = 0; d < 64; ++d) { q[d] = qp[d]; dO[d] = dop[d]; dq[d] = 0.f; delta = fmaf(dO[d], op[d], delta); }
  const float lse_i = lse[(int64_t)bh * S + i];
  for (int j = 0; j < S; ++j) {
    if (!mask[b * S + j]) continue;
    const float* kp = base + j * ld + H + h * 64;
    const float* vp = base + j * ld + 2 * H + h * 64;
    float s = 0.f, dpt = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) { s = fmaf(q[d], kp[d], s); dpt = fmaf(dO[d], vp[d], dpt); }
    const float p = expf(s * scale - lse_i);
    float pt = p, dp = dpt;
    if (drop.thr16) {
      const bool keep = nb_keep(drop, (uint32_t)((bh * S + i) * S + j));
      pt = keep ? p * drop.scale : 0.f;
      dp = keep ? dpt * drop.scale : 0.f;
    }
    const float ds = p * (dp - delta) * scale;
    (void)pt;
#pragma unroll
    for (int d = 0; d < 64; ++d) dq[d] = fmaf(ds, kp[d], dq[d]);
  }
  float* dqp = dbase + i * ld + h * 64;
#pragma unroll
  for (int d = 0; d < 64; ++d) dqp[d] = dq[d];
}

// dK / dV of the fp32 path: key j on the thread, queries in ascending order, every element of dK / dV written exactly once
// (masked keys: zero) - no memset, no atomics
__global__ __launch_bounds__(128) void attn_bwd_f32_kv_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                              const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                              const float* __restrict__ lse, float* __restrict__ dqkv, int S,
                                                              int heads, int H, float scale, DropCfg drop) {
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int j = blockIdx.y * blockDim.x + threadIdx.x;
  if (j >= S) return;
  const int64_t ld = 3 * (int64_t)H;
  const float* base = qkv + (int64_t)b * S * ld;
  float* dbase = dqkv + (int64_t)b * S * ld;
  const float* kp = base + j * ld + H + h * 64;
  const float* vp = base + j * ld + 2 * H + h * 64;
  float k[64], v[64], dk[64], dv[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) { k[d] = kp[d]; v[d] = vp[d]; dk[d] = 0.f; dv[d] = 0.f; }
  if (mask[b * S + j]) {
    for (int i = 0; i < S; ++i) {
      const float* qp = base + i * ld + h * 64;
      const float* dop = dctx + ((int64_t)b * S + i) * H + h * 64;
      const float* op = ctx + ((int64_t)b * S + i) * H + h * 64;
      float s = 0.f, dpt = 0.f, delta = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) { s = fmaf(qp[d], k[d], s); dpt = fmaf(dop[d], v[d], dpt); delta = fmaf(dop[d], op[d], delta); }
      const float p = expf(s * scale - lse[(int64_t)bh * S + i]);
      float pt = p, dp = dpt;
      if (drop.thr16) {
        const bool keep = nb_keep(drop, (uint32_t)((bh * S + i) * S + j));
        pt = keep ? p * drop.scale : 0.f;
        dp = keep ? dpt * drop.scale : 0.f;
      }
      const float ds = p * (dp - delta) * scale;
#pragma unroll
      for (int d = 0; d < 64; ++d) { dk[d] = fmaf(ds, qp[d], dk[d]); dv[d] = fmaf(pt, dop[d], dv[d]); }
    }
  }
  float* dkp = dbase + j * ld + H + h * 64;
  float* dvp = dbase + j * ld + 2 * H + h * 64;
#pragma unroll
  for (int d = 0; d < 64; ++d) { dkp[d] = dk[d]; dvp[d] = dv[d]; }
}

// =================================================================================================
// bf16 MFMA kernels
// =================================================================================================
// LDS-DMA a [Sp rows][64] bf16 tile (rows = tokens of one sequence, 64 columns starting at col_off)
__device__ __forceinline__ void stage_rows(__amdgpu_buffer_rsrc_t rs, char* tile, int Sp, int col_off, int ld, int tid, int nt = 256) {
  const int wave = tid >> 6;
  const int chunks = Sp * 8;                    // 16-byte chunks (a multiple of 256: whole waves are in or out)
  const int iters = (chunks + nt - 1) / nt;
  for (int i = 0; i < iters; ++i) {
    const int p = i * nt + tid;
    if (p >= chunks) break;
    const int row = p >> 3, slot = p & 7;
    const int c = slot ^ gsw(row);
    const uint32_t voff = (uint32_t)((row * ld + col_off + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (i * nt + wave * 64) * 16), 16, voff, 0, 0, 0);
  }
}

// ds_read_b128 row fragment of the 32x32x16 operand: lane l -> row (row0 + l&31), k = 16*ks + 8*(l>>5) + j
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31);
  const int c = 2 * ks + (lane >> 5);
  return *(const bf16x8*)(tile + row * 128 + ((c ^ gsw(row)) << 4));
}

// transposed (k-strided) fragment of the 32x32x16 B operand out of a [rows=k][64] tile:
// lane l -> column col0 + (l&31); PERMUTED = k order of an accumulator tile used as the A operand
// (element j of lane half hh is k = kbase + 8*(j>>2) + 4*hh + (j&3)); natural: k = kbase + 8*hh + j.
template <bool PERMUTED>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int kbase, int col0, int lane) {
  const int g4 = lane >> 4, hh = g4 >> 1, i = lane & 15, qq = i >> 2, pq = i & 3;
  const int col = col0 + 16 * (g4 & 1) + 4 * pq;
  const int r1 = kbase + (PERMUTED ? 4 * hh : 8 * hh) + qq;
  const int r2 = r1 + (PERMUTED ? 8 : 4);
  const char* a1 = tile + r1 * 128 + (((col >> 3) ^ gsw(r1)) << 4) + ((col & 4) ? 8 : 0);
  const char* a2 = tile + r2 * 128 + (((col >> 3) ^ gsw(r2)) << 4) + ((col & 4) ? 8 : 0);
  const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a1);
  const bf16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a2);
  bf16x8 o;
  o[0] = v1[0]; o[1] = v1[1]; o[2] = v1[2]; o[3] = v1[3]; o[4] = v2[0]; o[5] = v2[1]; o[6] = v2[2]; o[7] = v2[3];
  return o;
}

__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)a[8 * s + j];
  return o;
}

// write a wave's [32][64] fp32 accumulator pair (columns 0-31 / 32-63; lane = column) into rows
// row0.. of a plain [rows][64] bf16 LDS image, then store those 32 rows with 16-byte accesses.
__device__ __forceinline__ void store_tile(char* img, int row0, const f32x16& o0, const f32x16& o1, int lane, bf16* gbase,
                                           int64_t gld, int rows_valid, uint8_t* g8, float s8, float& amax,
                                           bool want_amax) {   // g8: optional e4m3 copy of (tile * s8); amax: running max |tile|
  // (amax by reference + flag: as an optional POINTER to a local it was address-taken and lived in scratch memory)
  const int hh = lane >> 5, c = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    char* rowp = img + (row0 + crow(r, hh)) * 128;
    *(bf16*)(rowp + c * 2) = (bf16)o0[r];
    *(bf16*)(rowp + 64 + c * 2) = (bf16)o1[r];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = i * 64 + lane, row = p >> 3, ch = p & 7;
    if (row < rows_valid) {
      const i32x4 v = *(const i32x4*)(img + (row0 + row) * 128 + ch * 16);
      if (gbase) *(i32x4*)(gbase + (int64_t)row * gld + ch * 8) = v;   // (null: only the e4m3 copy is wanted)
      if (g8 || want_amax) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, v);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)b[e];
        if (want_amax) {
#pragma unroll
          for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
        }
        if (g8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= s8;
          *(i32x2*)(g8 + (int64_t)row * gld + ch * 8) = i32x2{(int)fp8_pack4(f), (int)fp8_pack4(f + 4)};
        }
      }
    }
  }
}

__device__ __forceinline__ void store_tile(char* img, int row0, const f32x16& o0, const f32x16& o1, int lane, bf16* gbase,
                                           int64_t gld, int rows_valid, uint8_t* g8 = nullptr, float s8 = 1.f) {
  float unused = 0.f;
  store_tile(img, row0, o0, o1, lane, gbase, gld, rows_valid, g8, s8, unused, false);
}

// NKB <= 4 (S <= 128): every wave owns exactly one 32-query block, so (a) its Q fragments are fetched BEFORE the wait for the
// K / V tiles (one memory round trip less on the critical path of a 10-us workgroup) and (b) K is dead once the scores exist:
// after a barrier the per-wave output images reuse its rows, the workgroup needs 2 x Sp x 128 B (32.5 KiB at S = 128) instead
// of 48.5 KiB - four workgroups per CU instead of three, with the registers capped at 128 for that (launch bounds).  The kernel
// is bound by memory latency (0.8 MB in flight per workgroup, ~10 us each), not by arithmetic: residency is what it wants.
template <int NKB, bool DROP>
__global__ __launch_bounds__(256, (NKB <= 4 ? (DROP ? 3 : 4) : 1)) void attn_fwd_bf16_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                            bf16* __restrict__ ctx, float* __restrict__ lse, int S, int heads,
                                                            int H, float scale, DropCfg drop, uint8_t* __restrict__ ctx8,
                                                            uint32_t* __restrict__ keep_out, const uint32_t* __restrict__ a_prev,
                                                            uint32_t* __restrict__ a_new) {
  // ctx8 (fp8 forward): e4m3(ctx * s), s = 2^floor(log2(224 / amax)) from the amax ctx had in the previous pass (a_prev, delayed
  // scaling: common.h fp8_ascale_of); this pass's amax goes to a_new.  Both null: unit scale, nothing recorded.
  const float s8 = fp8_act_scale(a_prev);
  float amax8 = 0.f;
  // keep_out (optional, with dropout): the keep decisions as bit words [bh][key block][query] (bit j = key 32 kb + j kept), so
  // that the backward pass reads one word per (query, key block) instead of hashing every score element again
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int Sp = NKB * 32;
  constexpr bool kOne = (NKB <= 4);           // one query block per wave
  char* Kt = lds;
  char* Vt = lds + Sp * 128;
  float* madd = (float*)(lds + 2 * Sp * 128);
  char* Ost = kOne ? Kt : lds + 2 * Sp * 128 + Sp * 4;  // 4 x 4 KiB, one [32][64] image per wave (kOne: over K, see above)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int ld = 3 * H;
  const bf16* base = qkv + (int64_t)b * S * ld;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(S * ld * 2), 0x00020000);
  auto load_q = [&](int qb, bf16x8* qf) {
    const int qrow = 32 * qb + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const i32x4 raw = (qrow < S) ? *(const i32x4*)(base + (int64_t)qrow * ld + h * 64 + 16 * ks + 8 * hh) : i32x4{0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8, raw);
    }
  };
  bf16x8 qf[4];
  if (kOne) load_q(wave, qf);
  stage_rows(rs, Kt, Sp, H + h * 64, ld, tid);
  stage_rows(rs, Vt, Sp, 2 * H + h * 64, ld, tid);
  for (int k = tid; k < Sp; k += 256) madd[k] = (k < S && mask[b * S + k]) ? 0.f : -INFINITY;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float sl2 = scale * 1.4426950408889634f;   // scores in units of log2: exp(x) = exp2(x log2 e), one multiply less per element

  for (int qb = wave; qb < (kOne ? wave + 1 : NKB); qb += 4) {
    const int q0 = 32 * qb, qrow = q0 + (lane & 31);
    const bool active = qb < NKB;            // kOne: the waves beyond the last query block still take part in the barrier below
    if (!kOne) load_q(qb, qf);
    f32x16 sc[NKB];
    if (active) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, 32 * kb, ks, lane), qf[ks], a, 0, 0, 0);
      sc[kb] = a;
    }
    }
    if (kOne) __syncthreads();               // every wave has read its K fragments: the K rows become the output images
    if (!active) continue;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 ma = *(const f32x4*)(madd + 32 * kb + 8 * r4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = sc[kb][4 * r4 + e] * sl2 + ma[e];
          sc[kb][4 * r4 + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxs = (mx == -INFINITY) ? 0.f : mx;
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(sc[kb][r] - mxs);
        sc[kb][r] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (hh == 0 && qrow < S) lse[(int64_t)bh * S + qrow] = mxs * 0.6931471805599453f + __logf(sum);
    // normalise (+ dropout) and round to bf16 at once: the probabilities are the A operand of the P.V MFMA (k order of an
    // accumulator tile, see acc_to_frag) - kept as packed bf16 they occupy half the registers of the fp32 scores
    bf16x8 pa[NKB][2];
    if (DROP && drop.thr16 && (S & 1) == 0) {
      // registers (2j, 2j+1) hold keys (k, k+1) with k even: with S even they are one element pair of the counter
      // stream, so one hash decides both
      // counter of the pair = (row base + key) / 2 with an even row base: its product with the hash's odd constant splits into a
      // per-lane term and a compile-time term per register - no 32-bit multiply (quarter rate) per pair for the index
      const uint32_t rowbase = (uint32_t)((bh * S + qrow) * S);
      const uint32_t hb = ((rowbase >> 1) + 2u * (uint32_t)hh) * 0x9E3779B9U + drop.key;
      const float ids = inv * drop.scale;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        uint32_t kw = 0;          // this lane's 16 keys of the block, at the bit positions of lane half 0 (crow(r, 0)); shifted by 4 hh below
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t hsh = nb_hash32(hb + (uint32_t)((32 * kb + crow(r, 0)) >> 1) * 0x9E3779B9U);
          const bool k0 = (hsh & 0xFFFFu) >= drop.thr16, k1 = (hsh >> 16) >= drop.thr16;
          const float p0 = k0 ? sc[kb][r] * ids : 0.f;
          const float p1 = k1 ? sc[kb][r + 1] * ids : 0.f;
          kw |= (k0 ? (1u << crow(r, 0)) : 0u) | (k1 ? (2u << crow(r, 0)) : 0u);
          pa[kb][r >> 3][r & 7] = (bf16)p0;
          pa[kb][r >> 3][(r & 7) + 1] = (bf16)p1;
        }
        if (keep_out) {
          kw <<= 4 * hh;
          kw |= (uint32_t)__shfl_xor((int)kw, 32, 64);
          if (hh == 0 && qrow < Sp) keep_out[((int64_t)bh * NKB + kb) * Sp + qrow] = kw;
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        uint32_t kw = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = sc[kb][r] * inv;
          if (DROP && drop.thr16) {
            const int key = 32 * kb + crow(r, hh);
            const bool kp = nb_keep(drop, (uint32_t)((bh * S + qrow) * S + key));
            pv = kp ? pv * drop.scale : 0.f;
            kw |= kp ? (1u << crow(r, hh)) : 0u;
          }
          pa[kb][r >> 3][r & 7] = (bf16)pv;
        }
        if (DROP && keep_out && drop.thr16) {
          kw |= (uint32_t)__shfl_xor((int)kw, 32, 64);
          if (hh == 0 && qrow < Sp) keep_out[((int64_t)bh * NKB + kb) * Sp + qrow] = kw;
        }
      }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[kb][s], tr_frag<true>(Vt, 32 * kb + 16 * s, 0, lane), o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[kb][s], tr_frag<true>(Vt, 32 * kb + 16 * s, 32, lane), o1, 0, 0, 0);
      }
    store_tile(Ost + wave * 4096, 0, o0, o1, lane, ctx + ((int64_t)b * S + q0) * H + h * 64, H, S - q0,
               ctx8 ? ctx8 + ((int64_t)b * S + q0) * H + h * 64 : nullptr, s8, amax8, a_new != nullptr);
  }
  if (a_new) {
    amax8 = wave_max(amax8);
    if (lane == 0) amax_update(a_new, amax8);
  }
}

#ifdef NBEST_EXPERIMENTS   // first-generation backward: kept for A/B measurements only
// backward, Sp = 32*NKB <= 128.  LDS: Qt | Kt | Vt | dOt ([Sp][64] bf16 each) | dSb [Sp][128] bf16 | lse | delta | madd
template <int NKB>
__global__ __launch_bounds__(256) void attn_bwd_bf16_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                            const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
                                                            const float* __restrict__ lse, bf16* __restrict__ dqkv,
                                                            float* __restrict__ colpart, int S, int heads, int H,
                                                            float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int Sp = NKB * 32;
  char* Qt = lds;
  char* Kt = Qt + Sp * 128;
  char* Vt = Kt + Sp * 128;
  char* dOt = Vt + Sp * 128;
  char* dSb = dOt + Sp * 128;                    // [Sp][256 B], chunk ^= row & 15
  float* lse_s = (float*)(dSb + Sp * 256);
  float* del_s = lse_s + Sp;
  float* madd = del_s + Sp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int ld = 3 * H;
  const bf16* base = qkv + (int64_t)b * S * ld;
  const bf16* dobase = dctx + (int64_t)b * S * H;
  const bf16* obase = ctx + (int64_t)b * S * H;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(S * ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dobase, 0, (uint32_t)(S * H * 2), 0x00020000);
  stage_rows(rs, Qt, Sp, h * 64, ld, tid);
  stage_rows(rs, Kt, Sp, H + h * 64, ld, tid);
  stage_rows(rs, Vt, Sp, 2 * H + h * 64, ld, tid);
  stage_rows(rsd, dOt, Sp, h * 64, H, tid);
  for (int k = tid; k < Sp; k += 256) {
    madd[k] = (k < S && mask[b * S + k]) ? 0.f : -INFINITY;
    lse_s[k] = (k < S) ? lse[(int64_t)bh * S + k] : INFINITY;
  }
  {  // delta[q] = sum_d dO[q][d] * O[q][d]; two threads per row (Sp <= 128 -> 256 threads cover it)
    const int r = tid >> 1, half = tid & 1;
    float s = 0.f;
    if (r < S) {
      const bf16* dp = dobase + (int64_t)r * H + h * 64 + 32 * half;
      const bf16* op = obase + (int64_t)r * H + h * 64 + 32 * half;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a[8], o[8];
        Vec8<bf16>::load(dp + 8 * c, a);
        Vec8<bf16>::load(op + 8 * c, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(a[j], o[j], s);
      }
    }
    s += __shfl_xor(s, 1, 64);
    if (half == 0 && r < Sp) del_s[r] = s;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.f;
  if (wave < NKB) {
    const int key = 32 * wave + (lane & 31);
    const float mk = madd[key];
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = row_frag(Kt, 32 * wave, ks, lane);
      vf[ks] = row_frag(Vt, 32 * wave, ks, lane);
    }
#pragma unroll 1
    for (int qb = 0; qb < NKB; ++qb) {
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = da[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qt, 32 * qb, ks, lane), kf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dOt, 32 * qb, ks, lane), vf[ks], da, 0, 0, 0);
      }
      // sa/da: lane holds [q = 32qb + crow(r,hh)][key]
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 l4 = *(const f32x4*)(lse_s + 32 * qb + 8 * r4 + 4 * hh);
        const f32x4 d4 = *(const f32x4*)(del_s + 32 * qb + 8 * r4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * r4 + e;
          const int q = 32 * qb + 8 * r4 + 4 * hh + e;
          const float p = __expf(sa[r] * scale + mk - l4[e]);
          float pt = p, dp = da[r];
          if (drop.thr16) {
            const bool keep = nb_keep(drop, (uint32_t)((bh * S + q) * S + key));
            pt = keep ? p * drop.scale : 0.f;
            dp = keep ? dp * drop.scale : 0.f;
          }
          const float ds = p * (dp - d4[e]) * scale;
          sa[r] = pt;   // P~  (-> dV)
          da[r] = ds;   // dS' (-> dK, dQ)
          *(bf16*)(dSb + q * 256 + ((((key >> 3) ^ (q & 15))) << 4) + (key & 7) * 2) = (bf16)ds;
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pa = acc_to_frag(sa, s), dsa = acc_to_frag(da, s);
        const bf16x8 do0 = tr_frag<true>(dOt, 32 * qb + 16 * s, 0, lane), do1 = tr_frag<true>(dOt, 32 * qb + 16 * s, 32, lane);
        const bf16x8 q0f = tr_frag<true>(Qt, 32 * qb + 16 * s, 0, lane), q1f = tr_frag<true>(Qt, 32 * qb + 16 * s, 32, lane);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, do0, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, do1, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, q0f, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, q1f, dk1, 0, 0, 0);
      }
    }
  }
  __syncthreads();  // dSb complete
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq0[r] = dq1[r] = 0.f;
  if (wave < NKB) {
#pragma unroll
    for (int ks = 0; ks < 2 * NKB; ++ks) {
      const int row = 32 * wave + (lane & 31);
      const int c = 2 * ks + hh;
      const bf16x8 dsf = *(const bf16x8*)(dSb + row * 256 + ((c ^ (row & 15)) << 4));
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, tr_frag<false>(Kt, 16 * ks, 0, lane), dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, tr_frag<false>(Kt, 16 * ks, 32, lane), dq1, 0, 0, 0);
    }
  }
  __syncthreads();  // everyone is done reading Qt / Kt / Vt / dSb: reuse them as output images / scratch
  if (colpart) {
    // fused Q|K|V bias gradient: column sums of this (sample, head)'s dQ, dK, dV -> colpart[b][3H]
    float* cs = (float*)dSb;   // [4 waves][3][64]
    const f32x16* tiles[6] = {&dq0, &dq1, &dk0, &dk1, &dv0, &dv1};
#pragma unroll
    for (int t6 = 0; t6 < 6; ++t6) {
      float x = 0.f;
      if (wave < NKB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x += (*tiles[t6])[r];
      }
      x += __shfl_xor(x, 32, 64);
      if (lane < 32) cs[(wave * 3 + (t6 >> 1)) * 64 + (t6 & 1) * 32 + lane] = x;
    }
    __syncthreads();
    if (tid < 192) {
      const int which = tid >> 6, dcol = tid & 63;
      const float x = (cs[(0 * 3 + which) * 64 + dcol] + cs[(1 * 3 + which) * 64 + dcol]) +
                      (cs[(2 * 3 + which) * 64 + dcol] + cs[(3 * 3 + which) * 64 + dcol]);
      colpart[(int64_t)b * 3 * H + which * H + h * 64 + dcol] = x;
    }
  }
  if (wave < NKB) {
    const int r0 = 32 * wave;
    bf16* g = dqkv + ((int64_t)b * S + r0) * ld + h * 64;
    store_tile(Qt, r0, dq0, dq1, lane, g, ld, S - r0);
    store_tile(Kt, r0, dk0, dk1, lane, g + H, ld, S - r0);
    store_tile(Vt, r0, dv0, dv1, lane, g + 2 * H, ld, S - r0);
  }
}

#endif  // NBEST_EXPERIMENTS

// ---- backward, second structure: S <= 256, small LDS footprint -----------------------------------------
// One wave per 32-key block (4 waves for S <= 128, 8 for S <= 256).  LDS: Qt | Kt | dOt ([Sp][64] bf16), the dS image of TWO query
// blocks and small per-row arrays; V fragments come straight from HBM into registers (row fragments are 16 contiguous bytes per
// lane).  After the barrier that completes the dS image of query block qb, wave (qb mod NW) turns it into dQ[qb] = dS . K while
// everybody proceeds with qb+1.  67 KiB at S = 128 -> two workgroups per CU; 136 KiB at S = 256.
// Round 3 (the kernel ran at 47 vector instructions per score element, two thirds of them bookkeeping):
//   * dS crosses LDS TRANSPOSED: a lane (= key) holds four consecutive queries per register quad, so it stores them as one 8-byte
//     row piece of a [key][64 q] image (the two 32-query halves of a 128-byte row double-buffer even / odd query blocks) instead of
//     sixteen 2-byte scatters into a [q][key] slab; the dQ product reads the image with the hardware-transposed reads it already
//     uses for K (same swizzle, same helper);
//   * KB: the dropout decisions come from the forward's keep words ([bh][key block][query], one bit per key; 2 KiB per head in
//     LDS): one AND per element instead of a three-multiply hash;
//   * exponent in units of log2 (lse pre-multiplied in the prologue): subtract, fma, v_exp.
template <int NKB, bool KB>
__global__ __launch_bounds__((NKB <= 4 ? 4 : 8) * 64, 2) void attn_bwd2_bf16_kernel(
    const bf16* __restrict__ qkv, const uint8_t* __restrict__ mask, const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
    const float* __restrict__ lse, bf16* __restrict__ dqkv, float* __restrict__ colpart, int S, int heads, int H, float scale,
    DropCfg drop, Fp8Grad f8, const uint32_t* __restrict__ keep) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int Sp = NKB * 32;
  constexpr int NW = (NKB <= 4) ? 4 : 8, NT = NW * 64;
  char* Qt = lds;
  char* Kt = Qt + Sp * 128;
  char* dOt = Kt + Sp * 128;
  char* dST = dOt + Sp * 128;                          // [Sp keys][64 q] bf16, chunk ^= gsw(key) like every [rows][64] tile
  float* lse_s = (float*)(dST + Sp * 128);             // lse * log2(e)
  float* del_s = lse_s + Sp;
  uint32_t* kw_s = (uint32_t*)(del_s + Sp);            // KB: [NKB key blocks][Sp queries] keep words of this head
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int ld = 3 * H;
  const bf16* base = qkv + (int64_t)b * S * ld;
  const bf16* dobase = dctx + (int64_t)b * S * H;
  const bf16* obase = ctx + (int64_t)b * S * H;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(S * ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dobase, 0, (uint32_t)(S * H * 2), 0x00020000);
  stage_rows(rs, Qt, Sp, h * 64, ld, tid, NT);
  stage_rows(rs, Kt, Sp, H + h * 64, ld, tid, NT);
  stage_rows(rsd, dOt, Sp, h * 64, H, tid, NT);
  for (int k = tid; k < Sp; k += NT) lse_s[k] = (k < S) ? lse[(int64_t)bh * S + k] * 1.4426950408889634f : INFINITY;
  if (KB) {
    for (int i = tid; i < NKB * Sp; i += NT) kw_s[i] = keep[(int64_t)bh * NKB * Sp + i];
  }
  {  // delta[q] = sum_d dO[q][d] * O[q][d]; two threads per row
    const int r = tid >> 1, half = tid & 1;
    float sdel = 0.f;
    if (r < S) {
      const bf16* dp = dobase + (int64_t)r * H + h * 64 + 32 * half;
      const bf16* op = obase + (int64_t)r * H + h * 64 + 32 * half;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a[8], o[8];
        Vec8<bf16>::load(dp + 8 * c, a);
        Vec8<bf16>::load(op + 8 * c, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) sdel = fmaf(a[j], o[j], sdel);
      }
    }
    sdel += __shfl_xor(sdel, 1, 64);
    if (half == 0 && r < Sp) del_s[r] = sdel;
  }
  // this wave's key block: mask bit and V row fragments straight from HBM
  const int key = 32 * wave + (lane & 31);
  const bool kvalid = (wave < NKB) && key < S;
  const float mk = (kvalid && mask[b * S + key]) ? 0.f : -INFINITY;
  const uint32_t lanebit = 1u << (lane & 31);
  const float sl2 = scale * 1.4426950408889634f;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const i32x4 raw = kvalid ? *(const i32x4*)(base + (int64_t)key * ld + 2 * H + h * 64 + 16 * ks + 8 * hh) : i32x4{0, 0, 0, 0};
    vf[ks] = __builtin_bit_cast(bf16x8, raw);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave < NKB) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = row_frag(Kt, 32 * wave, ks, lane);
  }
  f32x16 dk0, dk1, dv0, dv1, dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) dk0[r] = dk1[r] = dv0[r] = dv1[r] = dq0[r] = dq1[r] = 0.f;
  char* dsrow = dST + key * 128;                       // this lane's row of the dS image
  const int dsg = gsw(key);

#pragma unroll 1
  for (int qb = 0; qb < NKB; ++qb) {
    const int half = qb & 1;
    if (wave < NKB) {
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = da[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qt, 32 * qb, ks, lane), kf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dOt, 32 * qb, ks, lane), vf[ks], da, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int qoff = 32 * qb + 8 * r4 + 4 * hh;      // first of this lane's four consecutive queries
        const f32x4 l4 = *(const f32x4*)(lse_s + qoff);
        const f32x4 d4 = *(const f32x4*)(del_s + qoff);
        i32x4 k4 = {0, 0, 0, 0};
        if (KB) k4 = *(const i32x4*)(kw_s + wave * Sp + qoff);
        bf16x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * r4 + e;
          const float pv = __builtin_amdgcn_exp2f(fmaf(sa[r], sl2, mk - l4[e]));
          float pt = pv, dp = da[r];
          if (KB) {
            const float kf_ = ((uint32_t)k4[e] & lanebit) ? drop.scale : 0.f;
            pt = pv * kf_;
            dp = dp * kf_;
          } else if (drop.thr16) {
            const bool keep_ = nb_keep(drop, (uint32_t)((bh * S + qoff + e) * S + key));
            pt = keep_ ? pv * drop.scale : 0.f;
            dp = keep_ ? dp * drop.scale : 0.f;
          }
          const float ds = pv * (dp - d4[e]) * scale;
          sa[r] = pt;
          da[r] = ds;
          pk[e] = (bf16)ds;
        }
        const int col = 32 * half + 8 * r4 + 4 * hh;
        *(bf16x4*)(dsrow + (((col >> 3) ^ dsg) << 4) + ((col & 4) ? 8 : 0)) = pk;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pa = acc_to_frag(sa, s2), dsa = acc_to_frag(da, s2);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tr_frag<true>(dOt, 32 * qb + 16 * s2, 0, lane), dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tr_frag<true>(dOt, 32 * qb + 16 * s2, 32, lane), dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, tr_frag<true>(Qt, 32 * qb + 16 * s2, 0, lane), dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, tr_frag<true>(Qt, 32 * qb + 16 * s2, 32, lane), dk1, 0, 0, 0);
      }
    }
    __syncthreads();   // the dS image half of query block qb is complete (and the other half is free again)
    if (wave == (qb % NW)) {
#pragma unroll
      for (int ks = 0; ks < 2 * NKB; ++ks) {
        const bf16x8 dsf = tr_frag<false>(dST, 16 * ks, 32 * half, lane);       // dS[q = lane & 31][16 keys of step ks]
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, tr_frag<false>(Kt, 16 * ks, 0, lane), dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, tr_frag<false>(Kt, 16 * ks, 32, lane), dq1, 0, 0, 0);
      }
    }
  }
  __syncthreads();  // everyone is done with Qt / Kt / dOt / the dS image: reuse them as output images
  if (colpart) {
    float* cs = (float*)dST;   // [NW][3][64]
    const f32x16* tiles[6] = {&dq0, &dq1, &dk0, &dk1, &dv0, &dv1};
#pragma unroll
    for (int t6 = 0; t6 < 6; ++t6) {
      float x = 0.f;
      if (wave < NKB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x += (*tiles[t6])[r];
      }
      x += __shfl_xor(x, 32, 64);
      if (lane < 32) cs[(wave * 3 + (t6 >> 1)) * 64 + (t6 & 1) * 32 + lane] = x;
    }
    __syncthreads();
    if (tid < 192) {
      const int which = tid >> 6, dcol = tid & 63;
      float x = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) x += cs[(w2 * 3 + which) * 64 + dcol];
      colpart[(int64_t)b * 3 * H + which * H + h * 64 + dcol] = x;
    }
  }
  if (wave < NKB) {
    const int r0 = 32 * wave;      // wave w holds dQ of query block w (w = qb mod NW, NKB <= NW) and dK/dV of key block w
    const int64_t goff = ((int64_t)b * S + r0) * ld + h * 64;
    bf16* g = dqkv ? dqkv + goff : nullptr;
    uint8_t* g8 = f8.out8 ? f8.out8 + goff : nullptr;
    const float s8 = fp8_grad_scale(f8.amax_prev);
    float amax8 = 0.f;
    const bool am = f8.amax_new != nullptr;
    store_tile(Qt, r0, dq0, dq1, lane, g, ld, S - r0, g8, s8, amax8, am);
    store_tile(Kt, r0, dk0, dk1, lane, g ? g + H : nullptr, ld, S - r0, g8 ? g8 + H : nullptr, s8, amax8, am);
    store_tile(dOt, r0, dv0, dv1, lane, g ? g + 2 * H : nullptr, ld, S - r0, g8 ? g8 + 2 * H : nullptr, s8, amax8, am);
    if (f8.amax_new) {
      amax8 = wave_max(amax8);
      if (lane == 0) amax_update(f8.amax_new, amax8);
    }
  }
}


// ---- backward, third structure (round 4): 96 < S <= 128, EIGHT waves of 16 keys on v_mfma_f32_16x16x32_bf16 ------------------------
// tools/attn_scale.py: one workgroup of the second structure alone on a CU, operands cached, takes ~15 us for ~3 us of instruction issue
// per wave - a chain of dependent LDS read -> MFMA -> exp -> pack -> MFMA steps on ONE 32 x 32 tile per wave, at two waves per SIMD (223
// registers, 67 KiB of LDS).  Here a wave owns 16 keys: half the accumulators (dK, dV: 16 + 16 registers), half the elements per
// lane, and the dQ product of a query block is EIGHT 16 x 16 tiles - one per wave - instead of 16 MFMAs on one wave while the others
// wait at the barrier.  Same LDS images (Q | K | dO [128][64] bf16, chunk ^= gsw(row); dS image [key][64 q], two 32-query halves), same
// "key on the lane" arithmetic: S = Q K^T and dP = dO V^T tiles are D[16 q][16 keys] - a lane holds ONE key (lane & 15) and four
// consecutive queries per register quad; P^T and dS^T feed the dV / dK MFMAs as the A operand straight from the accumulators (k order
// = this lane group's queries {4g + e, 16 + 4g + e}; dO / Q fetched with the matching permuted transposed read).
__device__ __forceinline__ bf16x8 row_frag16(const char* tile, int row0, int ks, int lane) {   // rows row0 + (l & 15), k = 32 ks + 8 (l >> 4) + j
  const int row = row0 + (lane & 15);
  const int c = 4 * ks + (lane >> 4);
  return *(const bf16x8*)(tile + row * 128 + ((c ^ gsw(row)) << 4));
}
// operand with k along the image ROWS: lane l -> column col0 + (l & 15); natural: k = kbase + 8 (l >> 4) + j; PERMUTED: k = kbase + 4 (l >> 4) + j
// for j < 4 and kbase + 16 + 4 (l >> 4) + (j - 4) for j >= 4 (the query order of an accumulator pair used as the A operand)
template <bool PERMUTED>
__device__ __forceinline__ bf16x8 tr_frag16(const char* tile, int kbase, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pq = i & 3;
  const int col = col0 + 4 * pq;
  const int r1 = kbase + (PERMUTED ? 4 * g : 8 * g) + qq;
  const int r2 = r1 + (PERMUTED ? 16 : 4);
  const char* a1 = tile + r1 * 128 + (((col >> 3) ^ gsw(r1)) << 4) + ((col & 4) ? 8 : 0);
  const char* a2 = tile + r2 * 128 + (((col >> 3) ^ gsw(r2)) << 4) + ((col & 4) ? 8 : 0);
  const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a1);
  const bf16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a2);
  bf16x8 o;
  o[0] = v1[0]; o[1] = v1[1]; o[2] = v1[2]; o[3] = v1[3]; o[4] = v2[0]; o[5] = v2[1]; o[6] = v2[2]; o[7] = v2[3];
  return o;
}

// NKB = 4: 96 < S <= 128, 8 waves.  NKB = 8: 224 < S <= 256, 16 waves (one workgroup per CU: 136 KiB of LDS, four waves per SIMD); the dQ
// product of a query block is still eight tiles over all 256 keys - waves 0-7 take the even query blocks, waves 8-15 the odd ones.
template <int NKB, bool KB>
__global__ __launch_bounds__(NKB * 128, 4) void attn_bwd3_bf16_kernel(
    const bf16* __restrict__ qkv, const uint8_t* __restrict__ mask, const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
    const float* __restrict__ lse, bf16* __restrict__ dqkv, float* __restrict__ colpart, int S, int heads, int H, float scale,
    DropCfg drop, Fp8Grad f8, const uint32_t* __restrict__ keep) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(NKB == 4 || NKB == 8, "attn_bwd3: 8 or 16 waves of 16 keys");
  constexpr int Sp = NKB * 32, NW = 2 * NKB, NT = NW * 64;
  char* Qt = lds;
  char* Kt = Qt + Sp * 128;
  char* dOt = Kt + Sp * 128;
  char* dST = dOt + Sp * 128;
  float* lse_s = (float*)(dST + Sp * 128);
  float* del_s = lse_s + Sp;
  uint32_t* kw_s = (uint32_t*)(del_s + Sp);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int ld = 3 * H;
  const bf16* base = qkv + (int64_t)b * S * ld;
  const bf16* dobase = dctx + (int64_t)b * S * H;
  const bf16* obase = ctx + (int64_t)b * S * H;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(S * ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dobase, 0, (uint32_t)(S * H * 2), 0x00020000);
  stage_rows(rs, Qt, Sp, h * 64, ld, tid, NT);
  stage_rows(rs, Kt, Sp, H + h * 64, ld, tid, NT);
  stage_rows(rsd, dOt, Sp, h * 64, H, tid, NT);
  for (int k = tid; k < Sp; k += NT) lse_s[k] = (k < S) ? lse[(int64_t)bh * S + k] * 1.4426950408889634f : INFINITY;
  if (KB) {
    for (int i = tid; i < NKB * Sp; i += NT) kw_s[i] = keep[(int64_t)bh * NKB * Sp + i];
  }
  {  // delta[q] = sum_d dO[q][d] * O[q][d]; four threads per row
    const int r = tid >> 2, quarter = tid & 3;
    float sdel = 0.f;
    if (r < S) {
      const bf16* dp = dobase + (int64_t)r * H + h * 64 + 16 * quarter;
      const bf16* op = obase + (int64_t)r * H + h * 64 + 16 * quarter;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float a[8], o[8];
        Vec8<bf16>::load(dp + 8 * c, a);
        Vec8<bf16>::load(op + 8 * c, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) sdel = fmaf(a[j], o[j], sdel);
      }
    }
    sdel += __shfl_xor(sdel, 1, 64);
    sdel += __shfl_xor(sdel, 2, 64);
    if (quarter == 0 && r < Sp) del_s[r] = sdel;
  }
  // this wave's 16 keys: mask bit and V row fragments straight from HBM (B operand of dP = dO V^T: column = key, k = d)
  const int key = 16 * wave + c16;
  const bool kvalid = key < S;
  const float mk = (kvalid && mask[b * S + key]) ? 0.f : -INFINITY;
  const uint32_t lanebit = 1u << (key & 31);
  const float sl2 = scale * 1.4426950408889634f;
  bf16x8 kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const i32x4 raw = kvalid ? *(const i32x4*)(base + (int64_t)key * ld + 2 * H + h * 64 + 32 * ks + 8 * g) : i32x4{0, 0, 0, 0};
    vf[ks] = __builtin_bit_cast(bf16x8, raw);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) kf[ks] = row_frag16(Kt, 16 * wave, ks, lane);
  f32x4 dk[4], dv[4], dq[4];             // dq: this wave's tile of query blocks 0..3 (NKB = 4) / 2 i + khalf (NKB = 8)
#pragma unroll
  for (int t = 0; t < 4; ++t) { dk[t] = f32x4{0, 0, 0, 0}; dv[t] = f32x4{0, 0, 0, 0}; }
#pragma unroll
  for (int t = 0; t < 4; ++t) dq[t] = f32x4{0, 0, 0, 0};
  char* dsrow = dST + key * 128;
  const int dsg = gsw(key);
  const int qt_mine = (wave & 7) >> 2, dt_mine = wave & 3;    // this wave's 16 x 16 tile of every query block's dQ
  const int khalf = wave >> 3;                                // NKB = 8: 0 = this wave multiplies the even query blocks' dQ, 1 = the odd ones'

#pragma unroll
  for (int qb = 0; qb < NKB; ++qb) {
    const int half = qb & 1;
    f32x4 sa[2], da[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      sa[qt] = f32x4{0, 0, 0, 0}; da[qt] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        sa[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag16(Qt, 32 * qb + 16 * qt, ks, lane), kf[ks], sa[qt], 0, 0, 0);
        da[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag16(dOt, 32 * qb + 16 * qt, ks, lane), vf[ks], da[qt], 0, 0, 0);
      }
    }
    bf16x8 pa, dsa;      // P^T and dS^T as A operands: element j < 4 = query 4 g + j of tile 0, j >= 4 = query 16 + 4 g + (j - 4)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qoff = 32 * qb + 16 * qt + 4 * g;
      const f32x4 l4 = *(const f32x4*)(lse_s + qoff);
      const f32x4 d4 = *(const f32x4*)(del_s + qoff);
      i32x4 k4 = {0, 0, 0, 0};
      if (KB) k4 = *(const i32x4*)(kw_s + (wave >> 1) * Sp + qoff);
      bf16x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sa[qt][e], sl2, mk - l4[e]));
        float pt = pv, dp = da[qt][e];
        if (KB) {
          const float kf_ = ((uint32_t)k4[e] & lanebit) ? drop.scale : 0.f;
          pt = pv * kf_;
          dp = dp * kf_;
        } else if (drop.thr16) {
          const bool keep_ = nb_keep(drop, (uint32_t)((bh * S + qoff + e) * S + key));
          pt = keep_ ? pv * drop.scale : 0.f;
          dp = keep_ ? dp * drop.scale : 0.f;
        }
        const float ds = pv * (dp - d4[e]) * scale;
        pa[4 * qt + e] = (bf16)pt;
        dsa[4 * qt + e] = (bf16)ds;
        pk[e] = (bf16)ds;
      }
      const int col = 32 * half + 16 * qt + 4 * g;
      *(bf16x4*)(dsrow + (((col >> 3) ^ dsg) << 4) + ((col & 4) ? 8 : 0)) = pk;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, tr_frag16<true>(dOt, 32 * qb, 16 * dt, lane), dv[dt], 0, 0, 0);
      dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsa, tr_frag16<true>(Qt, 32 * qb, 16 * dt, lane), dk[dt], 0, 0, 0);
    }
    __syncthreads();   // the dS image half of query block qb is complete (and the other half is free again)
    if (NKB == 4 || (qb & 1) == khalf) {   // dQ[qb] tile (qt_mine, dt_mine) = dS[16 q][all keys] . K[keys][16 d]
      const int slot = (NKB == 4) ? qb : (qb >> 1);      // (qb is an unrolled loop counter: a register index at compile time)
#pragma unroll
      for (int ks = 0; ks < NKB; ++ks)
        dq[slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag16<false>(dST, 32 * ks, 32 * half + 16 * qt_mine, lane),
                                                           tr_frag16<false>(Kt, 32 * ks, 16 * dt_mine, lane), dq[slot], 0, 0, 0);
    }
  }
  __syncthreads();  // everyone is done with Qt / Kt / dOt / the dS image: reuse them as output images
  if (colpart) {    // bias gradient: column sums of dQ | dK | dV over the rows of this (sample, head), fixed order
    float* cs = (float*)dST;          // [NW waves][2][64] for dK, dV, then [NW waves][16] for dQ
    float* csq = cs + NW * 2 * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float xk = (dk[dt][0] + dk[dt][1]) + (dk[dt][2] + dk[dt][3]);
      float xv = (dv[dt][0] + dv[dt][1]) + (dv[dt][2] + dv[dt][3]);
      xk += __shfl_xor(xk, 16, 64); xk += __shfl_xor(xk, 32, 64);
      xv += __shfl_xor(xv, 16, 64); xv += __shfl_xor(xv, 32, 64);
      if (g == 0) { cs[(wave * 2 + 0) * 64 + 16 * dt + c16] = xk; cs[(wave * 2 + 1) * 64 + 16 * dt + c16] = xv; }
    }
    float xq = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) xq += (dq[t][0] + dq[t][1]) + (dq[t][2] + dq[t][3]);
    xq += __shfl_xor(xq, 16, 64); xq += __shfl_xor(xq, 32, 64);
    if (g == 0) csq[wave * 16 + c16] = xq;
    __syncthreads();
    if (tid < 192) {
      const int which = tid >> 6, dcol = tid & 63;
      float x = 0.f;
      if (which == 0) {                                     // the waves that hold a tile of this d tile: dt, dt + 4 (, dt + 8, dt + 12)
#pragma unroll
        for (int w2 = 0; w2 < NW / 4; ++w2) x += csq[((dcol >> 4) + 4 * w2) * 16 + (dcol & 15)];
      }
      else {
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) x += cs[(w2 * 2 + (which - 1)) * 64 + dcol];
      }
      colpart[(int64_t)b * 3 * H + which * H + h * 64 + dcol] = x;
    }
    __syncthreads();
  }
  // accumulators -> plain [row][64] bf16 images over Qt / Kt / dOt, then 16-byte row stores (and the optional e4m3 copies / amax)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 16 * wave + 4 * g + e;
      *(bf16*)(Kt + row * 128 + (16 * dt + c16) * 2) = (bf16)dk[dt][e];
      *(bf16*)(dOt + row * 128 + (16 * dt + c16) * 2) = (bf16)dv[dt][e];
    }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int qb = (NKB == 4) ? t : 2 * t + khalf;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *(bf16*)(Qt + (32 * qb + 16 * qt_mine + 4 * g + e) * 128 + (16 * dt_mine + c16) * 2) = (bf16)dq[t][e];
  }
  __syncthreads();
  {
    const int64_t goff = (int64_t)b * S * ld + h * 64;
    bf16* gq = dqkv ? dqkv + goff : nullptr;
    uint8_t* g8 = f8.out8 ? f8.out8 + goff : nullptr;
    const float s8 = fp8_grad_scale(f8.amax_prev);
    const bool am = f8.amax_new != nullptr;
    float amax8 = 0.f;
    const char* imgs[3] = {Qt, Kt, dOt};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int it = 0; it < (Sp * 8) / NT; ++it) {
        const int p = it * NT + tid, row = p >> 3, ch = p & 7;
        if (row < S) {
          const i32x4 v = *(const i32x4*)(imgs[t] + row * 128 + ch * 16);
          if (gq) *(i32x4*)(gq + (int64_t)row * ld + t * H + ch * 8) = v;
          if (g8 || am) {
            const bf16x8 bb = __builtin_bit_cast(bf16x8, v);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)bb[e];
            if (am) {
#pragma unroll
              for (int e = 0; e < 8; ++e) amax8 = fmaxf(amax8, fabsf(f[e]));
            }
            if (g8) {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= s8;
              *(i32x2*)(g8 + (int64_t)row * ld + t * H + ch * 8) = i32x2{(int)fp8_pack4(f), (int)fp8_pack4(f + 4)};
            }
          }
        }
      }
    }
    if (am) {
      amax8 = wave_max(amax8);
      if (lane == 0) amax_update(f8.amax_new, amax8);
    }
  }
}


// =================================================================================================
// long sequences, 256 < S <= 512 (BERT's position table ends at 512; the reference never truncates,
// /root/reference/utils/bert_xlnet_inputs.py:87-94).  Same tiles, operand tricks and LDS images as above, but
// nothing is sized by a compile-time S: the key-block count is a runtime argument.
//   forward : K and V of the head stay in LDS (2 x 64 KiB at S = 512, one workgroup per CU); a wave walks the key blocks
//             TWICE per 32-query block - pass 1 keeps only the running row max / sum-exp (online softmax), pass 2
//             recomputes the scores and feeds the normalised bf16 probabilities to the P.V MFMA - so no score tile has
//             to survive in registers across key blocks.
//   backward: 8 waves; the keys are processed in halves of 256 (wave = 32 keys of the current half, K half in LDS);
//             Q and dO are STREAMED through a double-buffered [32][64] LDS block per query block (LDS-DMA of block
//             qb+1 overlaps the math of block qb); dQ of query block qb is accumulated across the halves in the
//             registers of wave qb mod 8 (two blocks per wave at S = 512).
// =================================================================================================
// rows [row_off, row_off + nrows) of a sequence -> LDS rows 0.. (swizzle by the LDS row); nrows * 8 chunks, nt threads
__device__ __forceinline__ void stage_rows_off(__amdgpu_buffer_rsrc_t rs, char* tile, int nrows, int row_off, int col_off, int ld,
                                               int t, int nt) {
  const int chunks = nrows * 8;
  for (int p = t; p < chunks; p += nt) {
    const int row = p >> 3, slot = p & 7;
    const int c = slot ^ gsw(row);
    const uint32_t voff = (uint32_t)(((row_off + row) * ld + col_off + c * 8) * 2);
    // LDS-DMA writes lane-linear from a wave-uniform base: chunk p of this wave's 64 lands at base + lane * 16
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (p & ~63) * 16), 16, voff, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void attn_fwd_long_bf16_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                                 bf16* __restrict__ ctx, float* __restrict__ lse, int S, int nkb,
                                                                 int heads, int H, float scale, DropCfg drop, uint8_t* __restrict__ ctx8,
                                                                 const uint32_t* __restrict__ a_prev, uint32_t* __restrict__ a_new) {
  const float s8 = fp8_act_scale(a_prev);       // see attn_fwd_bf16_kernel
  float amax8 = 0.f;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int Sp = nkb * 32;
  char* Kt = lds;
  char* Vt = lds + Sp * 128;
  float* madd = (float*)(lds + 2 * Sp * 128);
  char* Ost = lds + 2 * Sp * 128 + Sp * 4;  // 4 x 4 KiB, one [32][64] image per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int ld = 3 * H;
  const bf16* base = qkv + (int64_t)b * S * ld;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(S * ld * 2), 0x00020000);
  stage_rows(rs, Kt, Sp, H + h * 64, ld, tid);
  stage_rows(rs, Vt, Sp, 2 * H + h * 64, ld, tid);
  for (int k = tid; k < Sp; k += 256) madd[k] = (k < S && mask[b * S + k]) ? 0.f : -INFINITY;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qb = wave; qb < nkb; qb += 4) {
    const int q0 = 32 * qb, qrow = q0 + (lane & 31);
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const i32x4 raw = (qrow < S) ? *(const i32x4*)(base + (int64_t)qrow * ld + h * 64 + 16 * ks + 8 * hh) : i32x4{0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8, raw);
    }
    // ---- pass 1: running max m and sum-exp l over this lane's half of the keys ----
    float m = -INFINITY, l = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, 32 * kb, ks, lane), qf[ks], a, 0, 0, 0);
      float bm = -INFINITY;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 ma = *(const f32x4*)(madd + 32 * kb + 8 * r4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = a[4 * r4 + e] * scale + ma[e];
          a[4 * r4 + e] = v;
          bm = fmaxf(bm, v);
        }
      }
      const float mn = fmaxf(m, bm);
      if (mn > -INFINITY) {
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) add += __expf(a[r] - mn);
        l = ((m > -INFINITY) ? l * __expf(m - mn) : 0.f) + add;
      }
      m = mn;
    }
    const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
    const float mx = fmaxf(m, mo);
    const float mxs = (mx == -INFINITY) ? 0.f : mx;
    const float sum = ((m > -INFINITY) ? l * __expf(m - mxs) : 0.f) + ((mo > -INFINITY) ? lo * __expf(mo - mxs) : 0.f);
    const float inv = 1.0f / sum;
    if (hh == 0 && qrow < S) lse[(int64_t)bh * S + qrow] = mxs + __logf(sum);
    // ---- pass 2: probabilities (recomputed) -> dropout -> P . V ----
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    const uint32_t rowbase = (uint32_t)((bh * S + qrow) * S);
    const bool pair_hash = drop.thr16 && (S & 1) == 0;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, 32 * kb, ks, lane), qf[ks], a, 0, 0, 0);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 ma = *(const f32x4*)(madd + 32 * kb + 8 * r4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * r4 + e] = __expf(a[4 * r4 + e] * scale + ma[e] - mxs) * inv;
      }
      if (pair_hash) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t idx = rowbase + (uint32_t)(32 * kb + crow(r, hh));
          const uint32_t hsh = nb_hash32((idx >> 1) * 0x9E3779B9U + drop.key);
          a[r] = ((hsh & 0xFFFFu) >= drop.thr16) ? a[r] * drop.scale : 0.f;
          a[r + 1] = ((hsh >> 16) >= drop.thr16) ? a[r + 1] * drop.scale : 0.f;
        }
      } else if (drop.thr16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = nb_keep(drop, rowbase + (uint32_t)(32 * kb + crow(r, hh))) ? a[r] * drop.scale : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pa = acc_to_frag(a, s);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tr_frag<true>(Vt, 32 * kb + 16 * s, 0, lane), o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tr_frag<true>(Vt, 32 * kb + 16 * s, 32, lane), o1, 0, 0, 0);
      }
    }
    store_tile(Ost + wave * 4096, 0, o0, o1, lane, ctx + ((int64_t)b * S + q0) * H + h * 64, H, S - q0,
               ctx8 ? ctx8 + ((int64_t)b * S + q0) * H + h * 64 : nullptr, s8, amax8, a_new != nullptr);
  }
  if (a_new) {
    amax8 = wave_max(amax8);
    if (lane == 0) amax_update(a_new, amax8);
  }
}

__global__ __launch_bounds__(512) void attn_bwd_long_bf16_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                                 const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
                                                                 const float* __restrict__ lse, bf16* __restrict__ dqkv,
                                                                 float* __restrict__ colpart, int S, int nkb, int heads, int H,
                                                                 float scale, DropCfg drop, Fp8Grad f8) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NW = 8, NT = 512, KH = 256, RS = 512;
  const float s8 = fp8_grad_scale(f8.amax_prev);
  float amax8 = 0.f;
  const bool am = f8.amax_new != nullptr;       // keys per half; dS slab row stride (256 keys x 2 B)
  const int Sp = nkb * 32;
  char* Qs = lds;                                       // [2][32][128 B]
  char* dOs = Qs + 2 * 4096;                            // [2][32][128 B]
  char* Kt = dOs + 2 * 4096;                            // [KH][128 B]; after a half: 8 x 4 KiB output images
  char* dSb = Kt + KH * 128;                            // [2][32][RS]
  float* lse_s = (float*)(dSb + 2 * 32 * RS);
  float* del_s = lse_s + Sp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int ld = 3 * H;
  const bf16* base = qkv + (int64_t)b * S * ld;
  const bf16* dobase = dctx + (int64_t)b * S * H;
  const bf16* obase = ctx + (int64_t)b * S * H;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(S * ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dobase, 0, (uint32_t)(S * H * 2), 0x00020000);
  for (int k = tid; k < Sp; k += NT) lse_s[k] = (k < S) ? lse[(int64_t)bh * S + k] : INFINITY;
  for (int r = tid >> 1; r < Sp; r += NT / 2) {   // delta[q] = sum_d dO[q][d] * O[q][d]; two threads per row
    const int half = tid & 1;
    float sdel = 0.f;
    if (r < S) {
      const bf16* dp = dobase + (int64_t)r * H + h * 64 + 32 * half;
      const bf16* op = obase + (int64_t)r * H + h * 64 + 32 * half;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a[8], o[8];
        Vec8<bf16>::load(dp + 8 * c, a);
        Vec8<bf16>::load(op + 8 * c, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) sdel = fmaf(a[j], o[j], sdel);
      }
    }
    sdel += __shfl_xor(sdel, 1, 64);
    if (half == 0) del_s[r] = sdel;
  }
  f32x16 dqa0, dqa1, dqb0, dqb1;                  // dQ of query blocks `wave` and `wave + 8`
  float ck0 = 0.f, ck1 = 0.f, cv0 = 0.f, cv1 = 0.f;   // column sums of dK / dV over the halves (bias gradient)
#pragma unroll
  for (int r = 0; r < 16; ++r) dqa0[r] = dqa1[r] = dqb0[r] = dqb1[r] = 0.f;
  const int nkh = (nkb + 7) / 8;

#pragma unroll 1
  for (int kh = 0; kh < nkh; ++kh) {
    const int key0 = KH * kh;
    const int nkb_h = (nkb - 8 * kh < 8) ? nkb - 8 * kh : 8;
    __syncthreads();                               // the previous half is done with Kt (output images), Qs / dOs and the slabs
    stage_rows_off(rs, Kt, KH, key0, H + h * 64, ld, tid, NT);
    if (wave < 4) stage_rows_off(rs, Qs, 32, 0, h * 64, ld, tid, 256);
    else stage_rows_off(rsd, dOs, 32, 0, h * 64, H, tid - 256, 256);
    const int keyl = 32 * wave + (lane & 31), key = key0 + keyl;
    const bool active = wave < nkb_h;
    const bool kvalid = active && key < S;
    const float mk = (kvalid && mask[b * S + key]) ? 0.f : -INFINITY;
    // K row fragments are re-read from LDS and V row fragments from L2 in every query block: holding them (32 registers)
    // next to two dQ tile pairs would spill
    const bf16* vrow = base + (int64_t)(kvalid ? key : 0) * ld + 2 * H + h * 64 + 8 * hh;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.f;

#pragma unroll 1
    for (int qb = 0; qb < nkb; ++qb) {
      const int cur = qb & 1;
      const char* Qc = Qs + cur * 4096;
      const char* dOc = dOs + cur * 4096;
      char* slab = dSb + cur * 32 * RS;
      if (qb + 1 < nkb) {                          // block qb+1 -> the buffers everyone finished reading before the last barrier
        if (wave < 4) stage_rows_off(rs, Qs + (cur ^ 1) * 4096, 32, 32 * (qb + 1), h * 64, ld, tid, 256);
        else stage_rows_off(rsd, dOs + (cur ^ 1) * 4096, 32, 32 * (qb + 1), h * 64, H, tid - 256, 256);
      }
      if (active) {
        f32x16 sa, da;
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = da[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const i32x4 raw = kvalid ? *(const i32x4*)(vrow + 16 * ks) : i32x4{0, 0, 0, 0};
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qc, 0, ks, lane), row_frag(Kt, 32 * wave, ks, lane), sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dOc, 0, ks, lane), __builtin_bit_cast(bf16x8, raw), da, 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 l4 = *(const f32x4*)(lse_s + 32 * qb + 8 * r4 + 4 * hh);
          const f32x4 d4 = *(const f32x4*)(del_s + 32 * qb + 8 * r4 + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * r4 + e;
            const int ql = 8 * r4 + 4 * hh + e;
            const int q = 32 * qb + ql;
            const float pv = __expf(sa[r] * scale + mk - l4[e]);
            float pt = pv, dp = da[r];
            if (drop.thr16) {
              const bool keep = nb_keep(drop, (uint32_t)((bh * S + q) * S + key));
              pt = keep ? pv * drop.scale : 0.f;
              dp = keep ? dp * drop.scale : 0.f;
            }
            const float ds = pv * (dp - d4[e]) * scale;
            sa[r] = pt;
            da[r] = ds;
            *(bf16*)(slab + ql * RS + ((((keyl >> 3) ^ (ql & 15))) << 4) + (keyl & 7) * 2) = (bf16)ds;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pa = acc_to_frag(sa, s2), dsa = acc_to_frag(da, s2);
          dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tr_frag<true>(dOc, 16 * s2, 0, lane), dv0, 0, 0, 0);
          dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, tr_frag<true>(dOc, 16 * s2, 32, lane), dv1, 0, 0, 0);
          dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, tr_frag<true>(Qc, 16 * s2, 0, lane), dk0, 0, 0, 0);
          dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, tr_frag<true>(Qc, 16 * s2, 32, lane), dk1, 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of block qb+1 has landed
      __syncthreads();                                    // slab of block qb complete; block qb+1 visible to everyone
      if (wave == (qb & 7)) {                             // dQ[qb] += dS[qb, this half] . K[this half]
        // accumulated straight into the owning register pair (a temporary pair would push the kernel into scratch)
#define NB_DQ_ACC(D0, D1)                                                                                      \
  _Pragma("unroll 1") for (int ks = 0; ks < 2 * nkb_h; ++ks) {                                                 \
    const int row = lane & 31;                                                                                 \
    const int c = 2 * ks + hh;                                                                                 \
    const bf16x8 dsf = *(const bf16x8*)(slab + row * RS + ((c ^ (row & 15)) << 4));                            \
    D0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, tr_frag<false>(Kt, 16 * ks, 0, lane), D0, 0, 0, 0);      \
    D1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, tr_frag<false>(Kt, 16 * ks, 32, lane), D1, 0, 0, 0);     \
  }
        if (qb < 8) { NB_DQ_ACC(dqa0, dqa1) } else { NB_DQ_ACC(dqb0, dqb1) }
#undef NB_DQ_ACC
      }
    }
    __syncthreads();   // the last dQ update has read Kt and its slab: Kt becomes 8 output images
    if (active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { ck0 += dk0[r]; ck1 += dk1[r]; cv0 += dv0[r]; cv1 += dv1[r]; }
      const int r0 = key0 + 32 * wave;
      const int64_t goff = ((int64_t)b * S + r0) * ld + h * 64;
      bf16* g = dqkv ? dqkv + goff : nullptr;
      uint8_t* g8 = f8.out8 ? f8.out8 + goff : nullptr;
      store_tile(Kt + wave * 4096, 0, dk0, dk1, lane, g ? g + H : nullptr, ld, S - r0, g8 ? g8 + H : nullptr, s8, amax8, am);
      store_tile(Kt + wave * 4096, 0, dv0, dv1, lane, g ? g + 2 * H : nullptr, ld, S - r0, g8 ? g8 + 2 * H : nullptr, s8, amax8, am);
    }
  }
  __syncthreads();
  if (colpart) {
    float* cs = (float*)dSb;   // [NW][3][64]
    float cq0 = 0.f, cq1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cq0 += dqa0[r] + dqb0[r]; cq1 += dqa1[r] + dqb1[r]; }
    float v6[6] = {cq0, cq1, ck0, ck1, cv0, cv1};
#pragma unroll
    for (int t6 = 0; t6 < 6; ++t6) {
      float x = v6[t6];
      x += __shfl_xor(x, 32, 64);
      if (lane < 32) cs[(wave * 3 + (t6 >> 1)) * 64 + (t6 & 1) * 32 + lane] = x;
    }
    __syncthreads();
    if (tid < 192) {
      const int which = tid >> 6, dcol = tid & 63;
      float x = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) x += cs[(w2 * 3 + which) * 64 + dcol];
      colpart[(int64_t)b * 3 * H + which * H + h * 64 + dcol] = x;
    }
  }
  // dQ: wave w holds query blocks w and w + 8
  if (wave < nkb) {
    const int r0 = 32 * wave;
    const int64_t goff = ((int64_t)b * S + r0) * ld + h * 64;
    store_tile(Kt + wave * 4096, 0, dqa0, dqa1, lane, dqkv ? dqkv + goff : nullptr, ld, S - r0, f8.out8 ? f8.out8 + goff : nullptr, s8, amax8, am);
  }
  if (wave + 8 < nkb) {
    const int r0 = 32 * (wave + 8);
    const int64_t goff = ((int64_t)b * S + r0) * ld + h * 64;
    store_tile(Kt + wave * 4096, 0, dqb0, dqb1, lane, dqkv ? dqkv + goff : nullptr, ld, S - r0, f8.out8 ? f8.out8 + goff : nullptr, s8, amax8, am);
  }
  if (f8.amax_new) {
    amax8 = wave_max(amax8);
    if (lane == 0) amax_update(f8.amax_new, amax8);
  }
}

static size_t bwd2_lds_bytes(int nkb) {
  const int Sp = nkb * 32;       // Q | K | dO | dS image ([Sp][64] bf16 each), lse, delta, keep words [nkb][Sp]
  return (size_t)Sp * 128 * 4 + (size_t)Sp * 8 + (size_t)nkb * Sp * 4;
}
template <int NKB>
static void launch_bwd2(const bf16* qkv, const uint8_t* mask, const bf16* ctx, const bf16* dctx, const float* lse, bf16* dqkv,
                        float* colpart, int B, int S, int heads, int H, float scale, DropCfg d, hipStream_t st, Fp8Grad f8,
                        const uint32_t* keep) {
  const size_t sm = bwd2_lds_bytes(NKB);
  if (keep && d.thr16) {      // dropout decisions from the forward's keep words
    (void)hipFuncSetAttribute((const void*)attn_bwd2_bf16_kernel<NKB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attn_bwd2_bf16_kernel<NKB, true><<<B * heads, (NKB <= 4 ? 4 : 8) * 64, sm, st>>>(qkv, mask, ctx, dctx, lse, dqkv, colpart, S, heads, H, scale, d, f8, keep);
  } else {                    // no dropout, or no keep words (stand-alone calls): the decisions are hashed again
    (void)hipFuncSetAttribute((const void*)attn_bwd2_bf16_kernel<NKB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attn_bwd2_bf16_kernel<NKB, false><<<B * heads, (NKB <= 4 ? 4 : 8) * 64, sm, st>>>(qkv, mask, ctx, dctx, lse, dqkv, colpart, S, heads, H, scale, d, f8, nullptr);
  }
}

// K | V | key-mask addends (| 4 output images; S <= 128 restages the output over the K rows, which are dead by then)
static size_t fwd_lds_bytes(int nkb) {
  const size_t kv = (size_t)nkb * 32 * 256, madd = (size_t)nkb * 32 * 4;
  return (nkb <= 4) ? kv + madd : kv + madd + 4 * 4096;
}
#ifdef NBEST_EXPERIMENTS
static size_t bwd_lds_bytes(int nkb) { return (size_t)nkb * 32 * (4 * 128 + 256) + (size_t)nkb * 32 * 12; }
#endif

template <int NKB>
static void launch_fwd(const bf16* qkv, const uint8_t* mask, bf16* ctx, float* lse, int B, int S, int heads, int H, float scale,
                       DropCfg d, hipStream_t st, uint8_t* ctx8, uint32_t* keep, const uint32_t* a_prev, uint32_t* a_new) {
  const size_t sm = fwd_lds_bytes(NKB);
  if (d.thr16) {     // the dropout-free instantiation (evaluation, parity runs) carries no hash and no keep words: 58 -> 47 us at S = 128
    (void)hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<NKB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attn_fwd_bf16_kernel<NKB, true><<<B * heads, 256, sm, st>>>(qkv, mask, ctx, lse, S, heads, H, scale, d, ctx8, keep, a_prev, a_new);
  } else {
    (void)hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<NKB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attn_fwd_bf16_kernel<NKB, false><<<B * heads, 256, sm, st>>>(qkv, mask, ctx, lse, S, heads, H, scale, d, ctx8, keep, a_prev, a_new);
  }
}
#ifdef NBEST_EXPERIMENTS
template <int NKB>
static void launch_bwd(const bf16* qkv, const uint8_t* mask, const bf16* ctx, const bf16* dctx, const float* lse, bf16* dqkv,
                       float* colpart, int B, int S, int heads, int H, float scale, DropCfg d, hipStream_t st) {
  const size_t sm = bwd_lds_bytes(NKB);
  (void)hipFuncSetAttribute((const void*)attn_bwd_bf16_kernel<NKB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  attn_bwd_bf16_kernel<NKB><<<B * heads, 256, sm, st>>>(qkv, mask, ctx, dctx, lse, dqkv, colpart, S, heads, H, scale, d);
}
#endif

static int check_common(const char* who, int B, int S, int heads, int d, int dtype) {
  NB_CHECK(B > 0 && S > 0 && heads > 0, NBEST_ERR_SHAPE, "%s: bad shape", who);
  NB_CHECK(d == 64, NBEST_ERR_SHAPE, "%s: head dimension %d not supported (64 only)", who, d);
  NB_CHECK(dtype == NBEST_F32 || dtype == NBEST_BF16, NBEST_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
  NB_CHECK((int64_t)B * heads * S * S < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "%s: B*heads*S*S overflows the dropout counter", who);
  return NBEST_OK;
}

}  // namespace

// ctx8 != NULL (bf16 only): also write the e4m3 copy of ctx that the fp8 attention-output GEMM reads
// keep (bf16, S <= 256, optional): receives the dropout keep words [B * heads][ceil(S / 32)][32 ceil(S / 32)] for the backward pass
// (nbest_internal_attention_keep_bytes)
size_t nbest_internal_attention_keep_bytes(int B, int S, int heads) {
  const size_t nkb = (size_t)(S + 31) / 32;
  return (S <= 256) ? (size_t)B * heads * nkb * nkb * 32 * sizeof(uint32_t) : 0;
}
int nbest_internal_attention_fwd8(const void* qkv, const uint8_t* key_mask, void* ctx, void* ctx8, float* lse, int B, int S, int heads,
                                  int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, nbest_stream_t stream,
                                  uint32_t* keep, const uint32_t* a_prev, uint32_t* a_new) {
  NB_CHECK(qkv && key_mask && ctx && lse, NBEST_ERR_ARG, "attention_fwd: null pointer");
  if (int e = check_common("attention_fwd", B, S, heads, d, dtype)) return e;
  hipStream_t st = (hipStream_t)stream;
  const int H = heads * d;
  const float scale = 1.0f / sqrtf((float)d);
  const DropCfg dc = make_drop(drop_p, seed, drop_stream);
  if (dtype == NBEST_F32) {
    NB_CHECK(S <= 512, NBEST_ERR_SHAPE, "attention_fwd(f32): S=%d > 512", S);
    attn_fwd_f32_kernel<<<dim3(B * heads, (S + 127) / 128), 128, 0, st>>>((const float*)qkv, key_mask, (float*)ctx, lse, S, heads, H, scale, dc);
  } else {
    NB_CHECK(S <= 512, NBEST_ERR_SHAPE, "attention_fwd(bf16): S=%d > 512", S);
    NB_CHECK((int64_t)S * 3 * H * 2 < ((int64_t)1 << 31), NBEST_ERR_SHAPE, "attention_fwd: sequence too large");
    const int nkb = (S + 31) / 32;
    if (nkb > 8) {
      const size_t sm = fwd_lds_bytes(nkb);
      (void)hipFuncSetAttribute((const void*)attn_fwd_long_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
      attn_fwd_long_bf16_kernel<<<B * heads, 256, sm, st>>>((const bf16*)qkv, key_mask, (bf16*)ctx, lse, S, nkb, heads, H, scale, dc, (uint8_t*)ctx8, a_prev, a_new);
      NB_LAUNCH_CHECK();
      return NBEST_OK;
    }
#define F(N) case N: launch_fwd<N>((const bf16*)qkv, key_mask, (bf16*)ctx, lse, B, S, heads, H, scale, dc, st, (uint8_t*)ctx8, keep, a_prev, a_new); break;
    switch (nkb) { F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) }
#undef F
  }
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_attention_fwd(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S, int heads,
                                   int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, nbest_stream_t stream) {
  return nbest_internal_attention_fwd8(qkv, key_mask, ctx, nullptr, lse, B, S, heads, d, dtype, drop_p, seed, drop_stream, stream, nullptr, nullptr, nullptr);
}

int nbest_internal_partial_rows_sum(const float* part, int nrows, int N, float* out, int accumulate, hipStream_t st);

extern "C" size_t nbest_attention_bwd_ws_bytes(int B, int S, int heads) {
  const size_t a = (size_t)B * 3 * heads * 64 * sizeof(float);
  const size_t b = nbest_rowred_ws_bytes((int64_t)B * S, (int64_t)3 * heads * 64);
  return a > b ? a : b;
}

// f8 (bf16 only): e4m3 copy (scaled by the previous pass's amax) + amax of dqkv for the fp8 QKV dgrad GEMM
int nbest_internal_attention_bwd8(const void* qkv, const uint8_t* key_mask, const void* ctx, const void* dctx, const float* lse,
                                  void* dqkv, float* dbias, int accumulate, void* ws, size_t ws_bytes, int B, int S, int heads,
                                  int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, nbest_stream_t stream, Fp8Grad f8,
                                  const uint32_t* keep) {
  NB_CHECK(qkv && key_mask && ctx && dctx && lse && (dqkv || (f8.out8 && dtype == NBEST_BF16)), NBEST_ERR_ARG, "attention_bwd: null pointer");
  if (int e = check_common("attention_bwd", B, S, heads, d, dtype)) return e;
  NB_CHECK(!dbias || (ws && ws_bytes >= nbest_attention_bwd_ws_bytes(B, S, heads)), NBEST_ERR_WORKSPACE,
           "attention_bwd: bias-gradient workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int H = heads * d;
  const float scale = 1.0f / sqrtf((float)d);
  const DropCfg dc = make_drop(drop_p, seed, drop_stream);
  if (dtype == NBEST_F32) {
    NB_CHECK(S <= 512, NBEST_ERR_SHAPE, "attention_bwd(f32): S=%d > 512", S);
    attn_bwd_f32_kernel<<<dim3(B * heads, (S + 127) / 128), 128, 0, st>>>((const float*)qkv, key_mask, (const float*)ctx,
                                                                          (const float*)dctx, lse, (float*)dqkv, S, heads, H, scale, dc);
    NB_LAUNCH_CHECK();
    attn_bwd_f32_kv_kernel<<<dim3(B * heads, (S + 127) / 128), 128, 0, st>>>((const float*)qkv, key_mask, (const float*)ctx,
                                                                             (const float*)dctx, lse, (float*)dqkv, S, heads, H, scale, dc);
    NB_LAUNCH_CHECK();
    if (dbias) return nbest_colsum(dqkv, dbias, (int64_t)B * S, 3 * H, 3 * H, NBEST_F32, accumulate, ws, ws_bytes, stream);
    return NBEST_OK;
  }
  NB_CHECK(S <= 512, NBEST_ERR_SHAPE, "attention_bwd(bf16): S=%d > 512", S);
  const int nkb = (S + 31) / 32;
  float* colpart = dbias ? (float*)ws : nullptr;
  if (nkb > 8) {
    const size_t sm = (size_t)2 * 4096 * 2 + 256 * 128 + 2 * 32 * 512 + (size_t)nkb * 32 * 8;
    (void)hipFuncSetAttribute((const void*)attn_bwd_long_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attn_bwd_long_bf16_kernel<<<B * heads, 512, sm, st>>>((const bf16*)qkv, key_mask, (const bf16*)ctx, (const bf16*)dctx, lse,
                                                         (bf16*)dqkv, colpart, S, nkb, heads, H, scale, dc, f8);
    NB_LAUNCH_CHECK();
    if (dbias) return nbest_internal_partial_rows_sum(colpart, B, 3 * H, dbias, accumulate, st);
    return NBEST_OK;
  }
#ifdef NBEST_EXPERIMENTS
  // experiment builds (`make diag`): NBEST_ATTN_BWD=1 selects the first structure (everything in LDS, S <= 128) for A/B runs
  static const int old_structure = [] { const char* e = getenv("NBEST_ATTN_BWD"); return (e && e[0] == '1') ? 1 : 0; }();
  if (old_structure && nkb <= 4) {
#define F(N) case N: launch_bwd<N>((const bf16*)qkv, key_mask, (const bf16*)ctx, (const bf16*)dctx, lse, (bf16*)dqkv, colpart, B, S, heads, H, scale, dc, st); break;
    switch (nkb) { F(1) F(2) F(3) F(4) }
#undef F
    NB_LAUNCH_CHECK();
    if (dbias) return nbest_internal_partial_rows_sum(colpart, B, 3 * H, dbias, accumulate, st);
    return NBEST_OK;
  }
#endif
  // 96 < S <= 128: 16-key waves (attn_bwd3_bf16_kernel).  The 16-wave form for 224 < S <= 256 is correct (tests) but measured 0.7 - 0.9 % SLOWER on the
  // S = 256 steps than the second structure (one workgroup per CU either way, 25 spilled registers): experiment builds only (NBEST_ATTN_BWD=3).
  bool third = (nkb == 4);
#ifdef NBEST_EXPERIMENTS
  {
    static const int forced = [] { const char* e = getenv("NBEST_ATTN_BWD"); return (e && (e[0] == '2' || e[0] == '3')) ? e[0] - '0' : 0; }();
    if (forced == 2) third = false;
    if (forced == 3 && nkb == 8) third = true;
  }
#endif
  if (third) {
    const size_t sm = bwd2_lds_bytes(nkb);
    const bool kb = keep && dc.thr16;
#define G(N, K_)                                                                                                                        \
    do {                                                                                                                                \
      (void)hipFuncSetAttribute((const void*)attn_bwd3_bf16_kernel<N, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);        \
      attn_bwd3_bf16_kernel<N, K_><<<B * heads, N * 128, sm, st>>>((const bf16*)qkv, key_mask, (const bf16*)ctx, (const bf16*)dctx, lse, \
                                                                   (bf16*)dqkv, colpart, S, heads, H, scale, dc, f8, K_ ? keep : nullptr); \
    } while (0)
    if (nkb == 4) { if (kb) G(4, true); else G(4, false); }
#ifdef NBEST_EXPERIMENTS
    else { if (kb) G(8, true); else G(8, false); }
#endif
#undef G
    NB_LAUNCH_CHECK();
    if (dbias) return nbest_internal_partial_rows_sum(colpart, B, 3 * H, dbias, accumulate, st);
    return NBEST_OK;
  }
#define F(N) case N: launch_bwd2<N>((const bf16*)qkv, key_mask, (const bf16*)ctx, (const bf16*)dctx, lse, (bf16*)dqkv, colpart, B, S, heads, H, scale, dc, st, f8, keep); break;
  switch (nkb) { F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) }
#undef F
  NB_LAUNCH_CHECK();
  if (dbias) return nbest_internal_partial_rows_sum(colpart, B, 3 * H, dbias, accumulate, st);
  return NBEST_OK;
}

extern "C" int nbest_attention_bwd(const void* qkv, const uint8_t* key_mask, const void* ctx, const void* dctx, const float* lse,
                                   void* dqkv, float* dbias, int accumulate, void* ws, size_t ws_bytes, int B, int S, int heads,
                                   int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, nbest_stream_t stream) {
  return nbest_internal_attention_bwd8(qkv, key_mask, ctx, dctx, lse, dqkv, dbias, accumulate, ws, ws_bytes, B, S, heads, d, dtype, drop_p,
                                       seed, drop_stream, stream, Fp8Grad{nullptr, nullptr, nullptr}, nullptr);
}

extern "C" size_t nbest_attention_keep_bytes(int B, int S, int heads) { return nbest_internal_attention_keep_bytes(B, S, heads); }
extern "C" int nbest_attention_fwd_keep(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S, int heads,
                                        int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, void* keep,
                                        nbest_stream_t stream) {
  const bool ok = keep && dtype == NBEST_BF16 && nbest_internal_attention_keep_bytes(B, S, heads) > 0;
  return nbest_internal_attention_fwd8(qkv, key_mask, ctx, nullptr, lse, B, S, heads, d, dtype, drop_p, seed, drop_stream, stream,
                                       ok ? (uint32_t*)keep : nullptr, nullptr, nullptr);
}
extern "C" int nbest_attention_bwd_keep(const void* qkv, const uint8_t* key_mask, const void* ctx, const void* dctx, const float* lse,
                                        void* dqkv, float* dbias, int accumulate, void* ws, size_t ws_bytes, int B, int S, int heads,
                                        int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, const void* keep,
                                        nbest_stream_t stream) {
  const bool ok = keep && dtype == NBEST_BF16 && nbest_internal_attention_keep_bytes(B, S, heads) > 0;
  return nbest_internal_attention_bwd8(qkv, key_mask, ctx, dctx, lse, dqkv, dbias, accumulate, ws, ws_bytes, B, S, heads, d, dtype, drop_p,
                                       seed, drop_stream, stream, Fp8Grad{nullptr, nullptr, nullptr}, ok ? (const uint32_t*)keep : nullptr);
}
