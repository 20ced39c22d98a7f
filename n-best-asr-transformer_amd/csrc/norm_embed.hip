// HBM-bound row kernels: LayerNorm fwd/bwd (K5), embedding gather+LN fwd/bwd (K1/K1b), column sums
// (bias gradients), CLS-gradient scatter, fp32->bf16 arena cast.
//
// Layout: activations [M][H] row-major.  One wave64 owns one row at a time and keeps it in
// registers (H/4 float4 chunks, chunk c on lane c%64) - a row is read from HBM exactly once per
// pass; statistics are wave-shuffle reductions in fp32.  Loads are 16 B (fp32) / 8 B (bf16) per lane.
// Column reductions (dgamma, dbeta, bias gradients) are deterministic: per-block partial rows in a
// workspace + a finalize kernel (no float atomics on the gradient of a parameter).
#include "common.h"
#include <stdlib.h>

#ifndef NBEST_EMB_TPC
#define NBEST_EMB_TPC 16
#endif

namespace {

constexpr int kMaxRowredBlocks = 512;
constexpr int kMaxLnBwdBlocks = 1024;   // the LayerNorm backward may use more, smaller row blocks (workspace is sized for it)
static inline int rowred_blocks(int64_t M) {
  int64_t b = (M + 31) / 32;
  if (b < 1) b = 1;
  if (b > kMaxRowredBlocks) b = kMaxRowredBlocks;
  return (int)b;
}

__device__ __forceinline__ float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// ------------------------------------------------------------------------------------------------
template <typename T, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int64_t M, int H, float eps) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, nvec = H >> 2;
  const float invH = 1.0f / (float)H;
  for (int64_t row = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * wpb) {
    const T* xr = x + row * H;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) { v[i] = Vec4<T>::load(xr + 4 * c); s += sum4(v[i]); } else v[i] = f32x4{0, 0, 0, 0};
    }
    const float mean = wave_sum(s) * invH;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) { f32x4 d = v[i] - mean; q += sum4(d * d); }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * invH + eps);
    T* yr = y + row * H;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 g = *(const f32x4*)(gamma + 4 * c), b = *(const f32x4*)(beta + 4 * c);
        Vec4<T>::store(yr + 4 * c, (v[i] - mean) * rstd * g + b);
      }
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// bf16 fast path of the LayerNorm forward (H == VPL*256): no bounds checks, gamma / beta in registers, the next row
// in flight while this one is reduced, DPP reductions.  Two-pass variance as the generic kernel.
__device__ __forceinline__ f32x2 unpack2(uint32_t w) {
  return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
}
__device__ __forceinline__ uint32_t pack2(f32x2 v) {
  bf16x2 o;
  o[0] = (bf16)v[0]; o[1] = (bf16)v[1];
  return __builtin_bit_cast(uint32_t, o);
}
template <int VPL>
__global__ __launch_bounds__(256) void ln_fwd_fast_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16* __restrict__ y,
                                                          float* __restrict__ stats, int64_t M, float eps,
                                                          uint8_t* __restrict__ y8,      // optional e4m3 copy of y (fp8 forward):
                                                          const uint32_t* __restrict__ a_prev, uint32_t* __restrict__ a_new) {
  // y8 = e4m3(y * s), s = 2^floor(log2(224 / amax)) from the amax y had in the previous pass (delayed scaling, common.h
  // fp8_ascale_of); this pass's amax goes to a_new.  Both null: unit scale.
  const float s8 = fp8_act_scale(a_prev);
  float amax8 = 0.f;
  constexpr int H = VPL * 256;
  constexpr float invH = 1.0f / (float)H;
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x2 gm[VPL][2], bt[VPL][2];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 g4 = *(const f32x4*)(gamma + 4 * (lane + 64 * i)), b4 = *(const f32x4*)(beta + 4 * (lane + 64 * i));
    gm[i][0] = f32x2{g4[0], g4[1]}; gm[i][1] = f32x2{g4[2], g4[3]};
    bt[i][0] = f32x2{b4[0], b4[1]}; bt[i][1] = f32x2{b4[2], b4[3]};
  }
  uint2 nx[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) nx[i] = *(const uint2*)(x + row * H + 4 * (lane + 64 * i));
  for (; row < M; row += stride) {
    uint2 cx[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) cx[i] = nx[i];
    const int64_t nrow = (row + stride < M) ? row + stride : row;
#pragma unroll
    for (int i = 0; i < VPL; ++i) nx[i] = *(const uint2*)(x + nrow * H + 4 * (lane + 64 * i));
    f32x2 v[VPL][2];
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i][0] = unpack2(cx[i].x); v[i][1] = unpack2(cx[i].y);
      ps += v[i][0]; ps += v[i][1];
    }
    const float mean = wave_sum_dpp(ps[0] + ps[1]) * invH;
    f32x2 pq = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i][0] = v[i][0] - mean; v[i][1] = v[i][1] - mean;
      pq += v[i][0] * v[i][0]; pq += v[i][1] * v[i][1];
    }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(pq[0] + pq[1]) * invH + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const f32x2 o0 = v[i][0] * rstd * gm[i][0] + bt[i][0], o1 = v[i][1] * rstd * gm[i][1] + bt[i][1];
      const uint2 pk = uint2{pack2(o0), pack2(o1)};
      *(uint2*)(y + row * H + 4 * (lane + 64 * i)) = pk;
      if (y8) {   // e4m3 of the bf16 value that was just stored (what a cast of y would give), times the tensor's scale
        const f32x2 r0 = unpack2(pk.x), r1 = unpack2(pk.y);
        if (a_new) amax8 = fmaxf(fmaxf(amax8, fmaxf(fabsf(r0[0]), fabsf(r0[1]))), fmaxf(fabsf(r1[0]), fabsf(r1[1])));
        const float q[4] = {r0[0] * s8, r0[1] * s8, r1[0] * s8, r1[1] * s8};
        *(uint32_t*)(y8 + row * H + 4 * (lane + 64 * i)) = fp8_pack4(q);
      }
    }
    if (lane == 0) *(f32x2*)(stats + 2 * row) = f32x2{mean, rstd};
  }
  if (y8 && a_new) {
    amax8 = wave_max(amax8);
    if (lane == 0) amax_update(a_new, amax8);
  }
}

// partials layout: part[k][blk][H], k = 0 dgamma, 1 dbeta, 2 dbias(sum of dx)
template <typename T, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     T* __restrict__ dx, T* __restrict__ dx_drop, float* __restrict__ part,
                                                     int64_t M, int H, int rows_per_block, int with_dbias, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [3][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, nvec = H >> 2;
  const float invH = 1.0f / (float)H;
  for (int i = threadIdx.x; i < 3 * H; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  f32x4 ag[VPL], ab[VPL], ad[VPL], gm[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    ag[i] = ab[i] = ad[i] = f32x4{0, 0, 0, 0};
    const int c = lane + 64 * i;
    gm[i] = (c < nvec) ? *(const f32x4*)(gamma + 4 * c) : f32x4{0, 0, 0, 0};
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  // software prefetch: the (packed) x / dy of the NEXT row are in flight while this row is reduced
  typename Vec4<T>::raw_t nx[VPL], nd[VPL];
  float nmean = 0.f, nrstd = 0.f;
  {
    const int64_t row = r0 + wave;
    if (row < r1) {
      nmean = stats[2 * row]; nrstd = stats[2 * row + 1];
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) { nx[i] = Vec4<T>::raw_load(x + row * H + 4 * c); nd[i] = Vec4<T>::raw_load(dy + row * H + 4 * c); }
      }
    }
  }
  for (int64_t row = r0 + wave; row < r1; row += wpb) {
    const float mean = nmean, rstd = nrstd;
    typename Vec4<T>::raw_t cx[VPL], cd[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { cx[i] = nx[i]; cd[i] = nd[i]; }
    const int64_t nrow = row + wpb;
    if (nrow < r1) {
      nmean = stats[2 * nrow]; nrstd = stats[2 * nrow + 1];
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) { nx[i] = Vec4<T>::raw_load(x + nrow * H + 4 * c); nd[i] = Vec4<T>::raw_load(dy + nrow * H + 4 * c); }
      }
    }
    f32x4 xh[VPL], g[VPL], d[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        xh[i] = (Vec4<T>::cvt(cx[i]) - mean) * rstd;
        d[i] = Vec4<T>::cvt(cd[i]);
        g[i] = d[i] * gm[i];
        s1 += sum4(g[i]);
        s2 += sum4(g[i] * xh[i]);
      } else {
        xh[i] = g[i] = d[i] = f32x4{0, 0, 0, 0};
      }
    }
    s1 = wave_sum(s1) * invH;
    s2 = wave_sum(s2) * invH;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 o = (g[i] - s1 - xh[i] * s2) * rstd;
        Vec4<T>::store(dx + row * H + 4 * c, o);
        ag[i] += d[i] * xh[i];
        ab[i] += d[i];
        if (drop.thr16) {  // gradient that flows into the dense layer under the (regenerated) dropout mask
          const uint32_t k = nb_keep4(drop, (uint32_t)(row * H + 4 * c));
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (k >> e & 1) ? o[e] * drop.scale : 0.f;
          Vec4<T>::store(dx_drop + row * H + 4 * c, o);
        }
        ad[i] += o;
      }
    }
  }
  // the waves add their column sums into the block's LDS row one after the other (plain read-modify-write:
  // LDS float atomics cost ~10 us per workgroup here)
  for (int w = 0; w < wpb; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
          *(f32x4*)&acc[4 * c] += ag[i];
          *(f32x4*)&acc[H + 4 * c] += ab[i];
          if (with_dbias) *(f32x4*)&acc[2 * H + 4 * c] += ad[i];
        }
      }
    }
    __syncthreads();
  }
  const int nblk = gridDim.x;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    part[((int64_t)0 * nblk + blockIdx.x) * H + i] = acc[i];
    part[((int64_t)1 * nblk + blockIdx.x) * H + i] = acc[H + i];
    if (with_dbias) part[((int64_t)2 * nblk + blockIdx.x) * H + i] = acc[2 * H + i];
  }
}

// ---- bf16 fast path of the LayerNorm backward: H == VPL*256, no bounds checks, no divergent branches ----------
// The generic kernel above spends ~500 VALU instructions per row (exec-mask branches around every vector, register
// shuffles, six LDS-crossbar shuffles per reduction) and was VALU-issue bound at two waves per SIMD (59 us for 200 MB).
// Here: packed-pair fp32 math, bf16 pairs unpacked with one shift / one mask, DPP reductions, the dropout hash input
// formed by addition from a per-row scalar base, compile-time DROP / DBIAS, conflict-free LDS accumulation.
__device__ __forceinline__ f32x2 unpack_bf16x2(uint32_t w) { return unpack2(w); }
__device__ __forceinline__ uint32_t pack_bf16x2(f32x2 v) { return pack2(v); }

template <int VPL, bool DROP, bool DBIAS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ln_bwd_fast_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          bf16* __restrict__ dx, bf16* __restrict__ dx_drop,
                                                          float* __restrict__ part, int64_t M, int rows_per_block, DropCfg drop,
                                                          Fp8Grad f8) {   // f8: e4m3 copy / amax of the dense-branch gradient
  constexpr int H = VPL * 256;
  extern __shared__ __attribute__((aligned(16))) float acc[];   // [wave][k][e][i][lane]: every wave parks its column sums here
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float s8 = fp8_grad_scale(f8.amax_prev);
  float amax8 = 0.f;
  f32x2 gm[VPL][2], ag[VPL][2], ab[VPL][2], ad[VPL][2];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 g4 = *(const f32x4*)(gamma + 4 * (lane + 64 * i));
    gm[i][0] = f32x2{g4[0], g4[1]}; gm[i][1] = f32x2{g4[2], g4[3]};
    ag[i][0] = ag[i][1] = ab[i][0] = ab[i][1] = ad[i][0] = ad[i][1] = f32x2{0.f, 0.f};
  }
  // rows are dealt round-robin over ALL waves of the grid (rows_per_block < 0: stride = -rows_per_block), so at any
  // moment the grid streams one contiguous window of the tensors instead of gridDim.x*4 separate ones
  const int64_t stride = rows_per_block < 0 ? -(int64_t)rows_per_block : WAVES;
  const int64_t r0 = rows_per_block < 0 ? (int64_t)blockIdx.x * WAVES : (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = rows_per_block < 0 ? M : ((r0 + rows_per_block < M) ? r0 + rows_per_block : M);
  constexpr float invH = 1.0f / (float)H;
  int64_t row = r0 + wave;
  if (row < r1) {
    uint2 nx[VPL], nd[VPL];
    float nmean = stats[2 * row], nrstd = stats[2 * row + 1];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      nx[i] = *(const uint2*)(x + row * H + 4 * (lane + 64 * i));
      nd[i] = *(const uint2*)(dy + row * H + 4 * (lane + 64 * i));
    }
    for (; row < r1; row += stride) {
      const float mean = nmean, rstd = nrstd;
      uint2 cx[VPL], cd[VPL];
#pragma unroll
      for (int i = 0; i < VPL; ++i) { cx[i] = nx[i]; cd[i] = nd[i]; }
      {  // next row (clamped: the last iteration re-reads its own row instead of branching)
        const int64_t nrow = (row + stride < r1) ? row + stride : row;
        nmean = stats[2 * nrow]; nrstd = stats[2 * nrow + 1];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          nx[i] = *(const uint2*)(x + nrow * H + 4 * (lane + 64 * i));
          nd[i] = *(const uint2*)(dy + nrow * H + 4 * (lane + 64 * i));
        }
      }
      const float mr = -mean * rstd;
      f32x2 xh[VPL][2], d[VPL][2], g[VPL][2];
      f32x2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        xh[i][0] = unpack_bf16x2(cx[i].x) * rstd + mr; xh[i][1] = unpack_bf16x2(cx[i].y) * rstd + mr;
        d[i][0] = unpack_bf16x2(cd[i].x); d[i][1] = unpack_bf16x2(cd[i].y);
        g[i][0] = d[i][0] * gm[i][0]; g[i][1] = d[i][1] * gm[i][1];
        p1 += g[i][0]; p1 += g[i][1];
        p2 += g[i][0] * xh[i][0]; p2 += g[i][1] * xh[i][1];
      }
      float s1 = p1[0] + p1[1], s2 = p2[0] + p2[1];
      wave_sum2(s1, s2);
      const float a1 = -s1 * invH * rstd, a2 = -s2 * invH * rstd;
      const uint32_t hbase = (uint32_t)((row * H) >> 1) * 0x9E3779B9U + drop.key;   // hash input of the row's first pair
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        f32x2 o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) o[h] = xh[i][h] * a2 + (g[i][h] * rstd + a1);
        const int64_t off = row * H + 4 * (lane + 64 * i);
        *(uint2*)(dx + off) = uint2{pack_bf16x2(o[0]), pack_bf16x2(o[1])};
#pragma unroll
        for (int h = 0; h < 2; ++h) { ag[i][h] += d[i][h] * xh[i][h]; ab[i][h] += d[i][h]; }
        if (DROP) {
          const uint32_t q = (uint32_t)(2 * (lane + 64 * i));   // pair index inside the row
          const uint32_t h0 = nb_hash32(hbase + q * 0x9E3779B9U), h1 = nb_hash32(hbase + (q + 1) * 0x9E3779B9U);
          o[0][0] = ((h0 & 0xFFFFu) >= drop.thr16) ? o[0][0] * drop.scale : 0.f;
          o[0][1] = ((h0 >> 16) >= drop.thr16) ? o[0][1] * drop.scale : 0.f;
          o[1][0] = ((h1 & 0xFFFFu) >= drop.thr16) ? o[1][0] * drop.scale : 0.f;
          o[1][1] = ((h1 >> 16) >= drop.thr16) ? o[1][1] * drop.scale : 0.f;
          if (dx_drop) *(uint2*)(dx_drop + off) = uint2{pack_bf16x2(o[0]), pack_bf16x2(o[1])};   // (null: only its e4m3 copy is wanted)
        }
        if (DBIAS) { ad[i][0] += o[0]; ad[i][1] += o[1]; }
        if (f8.amax_new) {   // o = the gradient the dense-layer dgrad / wgrad GEMMs read (after the dropout mask)
          amax8 = fmaxf(fmaxf(amax8, fmaxf(fabsf(o[0][0]), fabsf(o[0][1]))), fmaxf(fabsf(o[1][0]), fabsf(o[1][1])));
          if (f8.out8) {
            const float q[4] = {o[0][0] * s8, o[0][1] * s8, o[1][0] * s8, o[1][1] * s8};
            *(uint32_t*)(f8.out8 + off) = fp8_pack4(q);
          }
        }
      }
    }
  }
  if (f8.amax_new) {
    amax8 = wave_max(amax8);
    if (lane == 0) amax_update(f8.amax_new, amax8);
  }
  float* mine = acc + wave * 3 * H;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int a = (e * VPL + i) * 64 + lane;
      mine[a] = ag[i][e >> 1][e & 1];
      mine[H + a] = ab[i][e >> 1][e & 1];
      if (DBIAS) mine[2 * H + a] = ad[i][e >> 1][e & 1];
    }
  __syncthreads();
  const int nblk = gridDim.x;
  for (int col = threadIdx.x; col < H; col += WAVES * 64) {
    const int chunk = col >> 2, e = col & 3, a = (e * VPL + (chunk >> 6)) * 64 + (chunk & 63);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      t0 += acc[w * 3 * H + a]; t1 += acc[w * 3 * H + H + a];
      if (DBIAS) t2 += acc[w * 3 * H + 2 * H + a];
    }
    part[((int64_t)0 * nblk + blockIdx.x) * H + col] = t0;
    part[((int64_t)1 * nblk + blockIdx.x) * H + col] = t1;
    if (DBIAS) part[((int64_t)2 * nblk + blockIdx.x) * H + col] = t2;
  }
}

// out_k[col] (+)= sum_blk part[k][blk][col];  grid (ceil(N/64), nout), block (64,16)
struct RowredOut {
  float* out[3];
  int accumulate[3];
};
__global__ __launch_bounds__(1024) void rowred_finalize_kernel(const float* __restrict__ part, int nblk, int N, RowredOut o) {
  // block = 32 columns x 32 row lanes (twice the workgroups of a 64 x 16 block for the same N, half the rows per
  // thread, four independent partial sums per thread): this launch is pure latency, 2 - 5 MB read by a few dozen blocks
  __shared__ float sm[32][33];
  const int k = blockIdx.y;
  const int col = blockIdx.x * 32 + threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < N && o.out[k] != nullptr) {
    const float* p = part + (int64_t)k * nblk * N + col;
    int b = threadIdx.y;
    for (; b + 96 < nblk; b += 128) {
      s0 += p[(int64_t)b * N]; s1 += p[(int64_t)(b + 32) * N]; s2 += p[(int64_t)(b + 64) * N]; s3 += p[(int64_t)(b + 96) * N];
    }
    for (; b < nblk; b += 32) s0 += p[(int64_t)b * N];
  }
  sm[threadIdx.y][threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.y == 0 && col < N && o.out[k] != nullptr) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += sm[r][threadIdx.x];
    o.out[k][col] = o.accumulate[k] ? o.out[k][col] + s : s;
  }
}

// column sums of X[M][N] (ld): block = 64 column-float4 lanes x 4 row lanes; grid (ceil(N/256), nblk)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ part, int64_t M,
                                                     int64_t N, int64_t ld, int rows_per_block) {
  __shared__ f32x4 sm[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 256 + 4 * tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  f32x4 a = {0, 0, 0, 0};
  if (col < N)
    for (int64_t r = r0 + ty; r < r1; r += 4) a += Vec4<T>::load(X + r * ld + col);
  sm[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && col < N) {
    a = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
    *(f32x4*)(part + (int64_t)blockIdx.y * N + col) = a;
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T, int VPL>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg,
                                                        const int64_t* __restrict__ pos, const T* __restrict__ word,
                                                        const T* __restrict__ type, const T* __restrict__ ptab,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ out, float* __restrict__ stats, int64_t M, int H,
                                                        float eps, DropCfg drop) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, nvec = H >> 2;
  const float invH = 1.0f / (float)H;
  for (int64_t row = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * wpb) {
    const T* wr = word + ids[row] * H;
    const T* tr = type + (seg ? seg[row] : 0) * H;
    const T* pr = ptab + pos[row] * H;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        v[i] = (Vec4<T>::load(wr + 4 * c) + Vec4<T>::load(tr + 4 * c)) + Vec4<T>::load(pr + 4 * c);
        s += sum4(v[i]);
      } else v[i] = f32x4{0, 0, 0, 0};
    }
    const float mean = wave_sum(s) * invH;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) { f32x4 d = v[i] - mean; q += sum4(d * d); }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * invH + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 g = *(const f32x4*)(gamma + 4 * c), b = *(const f32x4*)(beta + 4 * c);
        f32x4 o = (v[i] - mean) * rstd * g + b;
        if (drop.thr16) {
          const uint32_t k = nb_keep4(drop, (uint32_t)(row * H + 4 * c));
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (k >> e & 1) ? o[e] * drop.scale : 0.f;
        }
        Vec4<T>::store(out + row * H + 4 * c, o);
      }
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// Embedding backward, deterministic (round 4; rounds 1-3 summed the word table with fp32 atomics: 150 us at the atomic rate and
// a step that was not bit-reproducible).  LayerNorm backward per token gives de = o (fp32, registers only); three kernels:
//   embed_bwd_pos_kernel   block (j, y) owns sequence position j for the samples b = y, y + Y, ...: every row it sees has the same
//                          position key (pos[b * S + j] == pos[j]: BERT's arange; for the RoBERTa family the same except padding rows,
//                          whose key is the padding row and carries no gradient), so the position / token-type 0, 1 / LayerNorm-parameter
//                          sums stay in registers and leave as ONE partial row per block and quantity (summed by embed_bwd_finalize_kernel
//                          in a fixed order).  Rows with another position key / token type >= 2 fall back to row atomics (no family of
//                          the path produces them).
//   embed_bwd_word_kernel  the word table as a SEGMENTED REDUCE over the tokens sorted by word id (`perm`: a stable argsort of ids,
//                          built by the data loader next to ids): wave c owns `tpc` consecutive sorted tokens, recomputes o for each
//                          and adds runs of equal id in registers, in sorted order.  A run that begins and ends inside the chunk is
//                          stored straight into its table row (one plain 16-byte store per lane, no atomics, no memset of the touched
//                          rows needed); a run that crosses a chunk boundary leaves a partial row (slot 0: the run began before the
//                          chunk; slot 1: it began in the chunk and continues past it).
//   embed_bwd_wordfix_kernel  one block per chunk; the chunk in which a boundary-crossing run BEGINS sums that run's partial rows in
//                          chunk order (fixed tree: wave w takes every 4th row, the four wave sums are added in wave order) and stores
//                          the table row.  Frequent ids ([SEP]: (n_best + 1) x B tokens) span up to M / tpc chunks.
// Same additions in the same order on every run: the gradient is bit-reproducible.  Both token kernels recompute o (two reads of
// dout and of the gathered table rows, 2 x 100 MB at B = 256, S = 128 - mostly out of the Infinity Cache the second time) instead of
// writing it (100 MB fp32 out, 100 MB back in).
template <typename T, int VPL>
struct EmbTok {
  typename Vec4<T>::raw_t w[VPL], t[VPL], p[VPL], d[VPL];
  float mean, rstd;
};
// FULL: H == 256 VPL (every lane owns VPL whole chunks): no exec-mask guards, so a group of token loads is one straight run of
// global_load instructions the compiler never has to wait on early
template <typename T, int VPL, bool FULL>
__device__ __forceinline__ void emb_tok_load(EmbTok<T, VPL>& r, int64_t row, int64_t id, int64_t sv, int64_t key, const T* __restrict__ word,
                                             const T* __restrict__ type, const T* __restrict__ ptab, const float* __restrict__ stats,
                                             const T* __restrict__ dout, int H, int lane, int nvec) {
  const T* wr = word + id * H;
  const T* tr = type + sv * H;
  const T* pr = ptab + key * H;
  const T* dr = dout + row * H;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nvec) {
      r.w[i] = Vec4<T>::raw_load(wr + 4 * c); r.t[i] = Vec4<T>::raw_load(tr + 4 * c);
      r.p[i] = Vec4<T>::raw_load(pr + 4 * c); r.d[i] = Vec4<T>::raw_load(dr + 4 * c);
    }
  }
  r.mean = stats[2 * row]; r.rstd = stats[2 * row + 1];
}
// o = d(embedding sum) of the token; d / xh = dropout-masked upstream gradient and the normalised row (LayerNorm-parameter sums)
template <typename T, int VPL, bool FULL>
__device__ __forceinline__ void emb_tok_grad(const EmbTok<T, VPL>& r, int64_t row, const f32x4* gm, f32x4* o, f32x4* d, f32x4* xh, int H,
                                             int lane, int nvec, float invH, const DropCfg& drop) {
  f32x4 g[VPL];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nvec) {
      const f32x4 e = (Vec4<T>::cvt(r.w[i]) + Vec4<T>::cvt(r.t[i])) + Vec4<T>::cvt(r.p[i]);
      xh[i] = (e - r.mean) * r.rstd;
      d[i] = Vec4<T>::cvt(r.d[i]);
      if (drop.thr16) {
        const uint32_t k = nb_keep4(drop, (uint32_t)(row * H + 4 * c));
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) d[i][e2] = (k >> e2 & 1) ? d[i][e2] * drop.scale : 0.f;
      }
      g[i] = d[i] * gm[i];
      s1 += sum4(g[i]);
      s2 += sum4(g[i] * xh[i]);
    } else {
      xh[i] = g[i] = d[i] = f32x4{0, 0, 0, 0};
    }
  }
  wave_sum2(s1, s2);
  s1 *= invH; s2 *= invH;
#pragma unroll
  for (int i = 0; i < VPL; ++i) o[i] = (g[i] - s1 - xh[i] * s2) * r.rstd;
}
constexpr int kEmbGroupWord = 4, kEmbGroupPos = 2;   // tokens whose rows a wave has in flight at once (the position kernel carries five accumulator rows)

template <typename T, int VPL, bool FULL>
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg,
                                                            const int64_t* __restrict__ pos, const T* __restrict__ word,
                                                            const T* __restrict__ type, const T* __restrict__ ptab,
                                                            const float* __restrict__ gamma, const float* __restrict__ stats,
                                                            const T* __restrict__ dout, float* __restrict__ dtype_tab,
                                                            float* __restrict__ dptab, float* __restrict__ part, int* __restrict__ fixlist,
                                                            int B, int S, int H, int64_t pos_pad_id, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [5][H] column sums (dgamma, dbeta, position, type 0, type 1)
  // (readfirstlane: the wave index is uniform, but only this tells the compiler - otherwise every token index below is a
  // "divergent" value and each readlane becomes a waterfall loop)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6, nvec = H >> 2;
  const float invH = 1.0f / (float)H;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) fixlist[0] = 0;   // work list of embed_bwd_word_kernel (launched next)
  for (int i = threadIdx.x; i < 5 * H; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  f32x4 ag[VPL], ab[VPL], gm[VPL], ap[VPL], t0[VPL], t1[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    ag[i] = ab[i] = ap[i] = t0[i] = t1[i] = f32x4{0, 0, 0, 0};
    const int c = lane + 64 * i;
    gm[i] = (c < nvec) ? *(const f32x4*)(gamma + 4 * c) : f32x4{0, 0, 0, 0};
  }
  const int j = blockIdx.x;
  const int64_t key0 = pos[j];
  const int bstep = gridDim.y * wpb, b0 = blockIdx.y + gridDim.y * wave;
  // this wave's tokens b0, b0 + bstep, ...: indices of up to 64 of them at a time, one per lane (no dependent scalar load per token)
  for (int base = b0; base < B; base += 64 * bstep) {
    const int bl = base + lane * bstep;
    const int64_t lrow = (int64_t)(bl < B ? bl : b0) * S + j;
    const int64_t lid = ids[lrow], lkey = pos[lrow], lsv = seg ? seg[lrow] : 0;
    int n = (B - base + bstep - 1) / bstep;
    if (n > 64) n = 64;
    auto bc = [&](int64_t v, int l) -> int64_t {
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)v >> 32), l);
      return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    // groups of kEmbGroupPos tokens: all their rows are requested before the first is reduced
    for (int g0 = 0; g0 < n; g0 += kEmbGroupPos) {
      EmbTok<T, VPL> r[kEmbGroupPos];
#pragma unroll
      for (int u = 0; u < kEmbGroupPos; ++u) {
        const int k = (g0 + u < n) ? g0 + u : n - 1;
        emb_tok_load<T, VPL, FULL>(r[u], bc(lrow, k), bc(lid, k), bc(lsv, k), bc(lkey, k), word, type, ptab, stats, dout, H, lane, nvec);
      }
#pragma unroll
      for (int u = 0; u < kEmbGroupPos; ++u) {
        const int k = g0 + u;
        if (k >= n) break;
        const int64_t row = bc(lrow, k), key = bc(lkey, k), sv = bc(lsv, k);
        f32x4 o[VPL], d[VPL], xh[VPL];
        emb_tok_grad<T, VPL, FULL>(r[u], row, gm, o, d, xh, H, lane, nvec, invH, drop);
        const bool own_key = (key == key0), t_is0 = (sv == 0), t_is1 = (sv == 1);   // wave-uniform
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          ag[i] += d[i] * xh[i];
          ab[i] += d[i];
          if (own_key) ap[i] += o[i];
          if (t_is0) t0[i] += o[i];
          if (t_is1) t1[i] += o[i];
        }
        if ((!own_key && key != pos_pad_id) || (!t_is0 && !t_is1)) {   // not reached by BERT / RoBERTa-family inputs (see above)
#pragma unroll
          for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (!own_key && key != pos_pad_id) atomicAdd(dptab + key * H + 4 * c + e, o[i][e]);
                if (!t_is0 && !t_is1) atomicAdd(dtype_tab + sv * H + 4 * c + e, o[i][e]);
              }
          }
        }
      }
    }
  }
  for (int w = 0; w < wpb; ++w) {   // waves take turns, in wave order (no LDS float atomics, see ln_bwd_kernel)
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
          *(f32x4*)&acc[4 * c] += ag[i];
          *(f32x4*)&acc[H + 4 * c] += ab[i];
          *(f32x4*)&acc[2 * H + 4 * c] += ap[i];
          *(f32x4*)&acc[3 * H + 4 * c] += t0[i];
          *(f32x4*)&acc[4 * H + 4 * c] += t1[i];
        }
      }
    }
    __syncthreads();
  }
  const int nblk = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
  for (int i = threadIdx.x; i < H; i += blockDim.x)
#pragma unroll
    for (int k = 0; k < 5; ++k) part[((int64_t)k * nblk + bid) * H + i] = acc[k * H + i];
}

// sums of the partial rows of embed_bwd_pos_kernel, fixed order.  grid (ceil(H / 32), 4 + ceil(S / 32)), block (32, 32):
//   y = 0, 1, 2, 3: dgamma, dbeta, token-type rows 0 / 1 = sum over all S * Y blocks (32 row lanes + an LDS tree);
//   y = 4 + g: thread row r owns position j = 32 g + r: position row pos[j] = sum over the Y blocks (j, 0), (j, 1), ...
// `accum_ln` / `accum_tab`: add to what the outputs hold.
__global__ __launch_bounds__(1024) void embed_bwd_finalize_kernel(const float* __restrict__ part, const int64_t* __restrict__ pos, int S, int Y,
                                                                  int H, int n_types, int64_t pos_pad_id, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, float* __restrict__ dtype_tab,
                                                                  float* __restrict__ dptab, int accum_ln, int accum_tab) {
  __shared__ float sm[32][33];
  const int nblk = S * Y, col = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y;
  if (y >= 4) {
    const int j = (y - 4) * 32 + threadIdx.y;
    if (j >= S || col >= H) return;
    const int64_t key = pos[j];
    if (key == pos_pad_id) return;
    const float* p = part + ((int64_t)2 * nblk + j) * H + col;      // blocks (j, 0), (j, 1), ...: bid = yy * S + j
    float s = 0.f;
    for (int yy = 0; yy < Y; ++yy) s += p[(int64_t)yy * S * H];
    float* o = dptab + key * H + col;
    *o = accum_tab ? *o + s : s;
    return;
  }
  float* outp = y == 0 ? dgamma : (y == 1 ? dbeta : (y == 2 ? dtype_tab : (n_types > 1 ? dtype_tab + H : nullptr)));
  const float* p = part + (int64_t)(y < 2 ? y : y + 1) * nblk * H + col;
  const int acc = y < 2 ? accum_ln : accum_tab;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < H && outp != nullptr) {
    int b = threadIdx.y;
    for (; b + 96 < nblk; b += 128) {
      s0 += p[(int64_t)b * H]; s1 += p[(int64_t)(b + 32) * H]; s2 += p[(int64_t)(b + 64) * H]; s3 += p[(int64_t)(b + 96) * H];
    }
    for (; b < nblk; b += 32) s0 += p[(int64_t)b * H];
  }
  sm[threadIdx.y][threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.y == 0 && col < H && outp != nullptr) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += sm[r][threadIdx.x];
    outp[col] = acc ? outp[col] + s : s;
  }
}

// sorted id of sorted position i (out of range: sentinels that equal no id and not each other)
__device__ __forceinline__ int64_t emb_sid(const int32_t* __restrict__ perm, const int64_t* __restrict__ ids, int64_t i, int64_t M) {
  return i < 0 ? -2 : (i >= M ? -3 : ids[perm[i]]);
}

template <typename T, int VPL, bool FULL>
__global__ __launch_bounds__(256) void embed_bwd_word_kernel(const int32_t* __restrict__ perm, const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ seg, const int64_t* __restrict__ pos,
                                                             const T* __restrict__ word, const T* __restrict__ type,
                                                             const T* __restrict__ ptab, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, const T* __restrict__ dout,
                                                             float* __restrict__ dword, float* __restrict__ wpart, int* __restrict__ fixlist,
                                                             int64_t M, int H, int tpc, int64_t word_pad_id, int accumulate, DropCfg drop) {
  const int lane = threadIdx.x & 63, nvec = H >> 2;
  const int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: see the pos kernel
  const int64_t i0 = c * tpc;
  if (i0 >= M) return;
  const int64_t i1 = (i0 + tpc < M) ? i0 + tpc : M;
  const float invH = 1.0f / (float)H;
  // lane l holds sorted position i0 - 1 + l: the chunk's token rows and ids (tpc <= 62) plus the id before and after the chunk
  const int n = (int)(i1 - i0);
  const int64_t li = i0 - 1 + lane;
  const bool lin = lane <= n + 1 && li >= 0 && li < M;
  const int64_t lrow = lin ? perm[li] : 0;
  const int64_t lid = lin ? ids[lrow] : (li < 0 ? -2 : -3);          // sentinels equal no id
  const int64_t lsv = (seg && lin) ? seg[lrow] : 0;
  const int64_t lkey = lin ? pos[lrow] : 0;
  auto bc = [&](int64_t v, int l) -> int64_t {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
  };
  if (bc(lid, 1) == word_pad_id && bc(lid, n) == word_pad_id) return;   // sorted: the whole chunk is padding (no gradient)
  f32x4 gm[VPL], acc[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int cc = lane + 64 * i;
    gm[i] = (cc < nvec) ? *(const f32x4*)(gamma + 4 * cc) : f32x4{0, 0, 0, 0};
    acc[i] = f32x4{0, 0, 0, 0};
  }
  bool run_began_here = true;                                // does the running sum hold the FIRST token of its id?
  int64_t run_id = -5;
  for (int g0 = 1; g0 <= n; g0 += kEmbGroupWord) {               // groups of kEmbGroupWord tokens: every row requested before the first is reduced
    EmbTok<T, VPL> r[kEmbGroupWord];
#pragma unroll
    for (int u = 0; u < kEmbGroupWord; ++u) {
      const int k = (g0 + u <= n) ? g0 + u : n;
      emb_tok_load<T, VPL, FULL>(r[u], bc(lrow, k), bc(lid, k), bc(lsv, k), bc(lkey, k), word, type, ptab, stats, dout, H, lane, nvec);
    }
#pragma unroll
    for (int u = 0; u < kEmbGroupWord; ++u) {
      const int k = g0 + u;
      if (k > n) break;
      const int64_t id = bc(lid, k), row = bc(lrow, k);
      if (id == word_pad_id) continue;                       // padding rows carry no gradient
      if (id != run_id) {                                    // a new run starts with this token
        run_id = id;
        run_began_here = (bc(lid, k - 1) != id);             // k = 1: compares with the id before the chunk
#pragma unroll
        for (int i = 0; i < VPL; ++i) acc[i] = f32x4{0, 0, 0, 0};
      }
      f32x4 o[VPL], d[VPL], xh[VPL];
      emb_tok_grad<T, VPL, FULL>(r[u], row, gm, o, d, xh, H, lane, nvec, invH, drop);
#pragma unroll
      for (int i = 0; i < VPL; ++i) acc[i] += o[i];
      const int64_t id_next = bc(lid, k + 1);                // k = n: the id after the chunk
      if (id_next != id) {                                   // the run ends with this token
        float* dst = run_began_here ? dword + id * H : wpart + (c * 2 + 0) * H;
        const bool add = run_began_here && accumulate;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          const int cc = lane + 64 * i;
          if (FULL || cc < nvec) *(f32x4*)(dst + 4 * cc) = add ? *(const f32x4*)(dst + 4 * cc) + acc[i] : acc[i];
        }
      } else if (k == n) {                                   // the chunk ends inside the run
        float* dst = wpart + (c * 2 + (run_began_here ? 1 : 0)) * H;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          const int cc = lane + 64 * i;
          if (FULL || cc < nvec) *(f32x4*)(dst + 4 * cc) = acc[i];
        }
        // the chunk in which a boundary-crossing run BEGINS owns its fix-up (order of the list: irrelevant - every entry is a
        // different table row, and the additions inside a row are ordered by chunk)
        if (run_began_here && lane == 0) fixlist[1 + atomicAdd(fixlist, 1)] = (int)c;
      }
    }
  }
}

// Fix-up of the runs that cross chunk boundaries: a fixed grid of blocks walks embed_bwd_word_kernel's work list (chunks in
// which such a run begins); the block adds the run's partial rows in chunk order and stores the table row.
__global__ __launch_bounds__(256) void embed_bwd_wordfix_kernel(const int32_t* __restrict__ perm, const int64_t* __restrict__ ids,
                                                                const float* __restrict__ wpart, const int* __restrict__ fixlist,
                                                                float* __restrict__ dword, int64_t M, int H, int tpc, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];       // [4][H]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nvec = H >> 2;
  int* first = (int*)sm;                                           // [4], before the row sums use the buffer
  const int count = fixlist[0];
  for (int item = blockIdx.x; item < count; item += gridDim.x) {
    const int64_t c = fixlist[1 + item];
    const int64_t i1 = (c + 1) * tpc;                              // < M: the run continues past the chunk
    const int64_t id = emb_sid(perm, ids, i1 - 1, M);
    // chunks c + 1 .. ce continue the run (slot 0 each); the run ends in chunk ce = the first chunk q > c whose end is the end of
    // the data or is followed by another id.  256 candidates are probed at once (a serial walk costs two dependent loads per
    // chunk: 73 us for the (n_best + 1) x B [SEP] tokens of a batch).
    int64_t ce = c + 1;
    for (int64_t base = c + 1;; base += 256) {
      const int64_t q = base + threadIdx.x;
      const int64_t e1 = ((q + 1) * tpc < M) ? (q + 1) * tpc : M;
      const bool ends = e1 >= M || emb_sid(perm, ids, e1, M) != id;
      const unsigned long long m = __ballot(ends);
      if (lane == 0) first[wave] = m ? (__ffsll((long long)m) - 1) + 64 * wave : 1 << 30;
      __syncthreads();
      const int f = min(min(first[0], first[1]), min(first[2], first[3]));
      __syncthreads();
      if (f < (1 << 30)) { ce = base + f; break; }
    }
    for (int cc = lane; cc < nvec; cc += 64) {
      f32x4 s = {0, 0, 0, 0};
      int64_t q = c + 1 + wave;
      for (; q + 28 <= ce; q += 32) {                              // eight rows of this wave in flight, added in chunk order
        f32x4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *(const f32x4*)(wpart + ((q + 4 * u) * 2) * H + 4 * cc);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
      }
      for (; q <= ce; q += 4) s += *(const f32x4*)(wpart + (q * 2) * H + 4 * cc);
      *(f32x4*)(sm + wave * H + 4 * cc) = s;
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < nvec; cc += blockDim.x) {
      f32x4 s = *(const f32x4*)(wpart + (c * 2 + 1) * H + 4 * cc);   // the part of the run inside its first chunk
      s += *(const f32x4*)(sm + 4 * cc); s += *(const f32x4*)(sm + H + 4 * cc);
      s += *(const f32x4*)(sm + 2 * H + 4 * cc); s += *(const f32x4*)(sm + 3 * H + 4 * cc);
      float* dst = dword + id * H + 4 * cc;
      *(f32x4*)dst = accumulate ? *(const f32x4*)dst + s : s;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void cls_scatter_kernel(const float* __restrict__ dcls, T* __restrict__ dh, int64_t nvec_total, int S, int H) {
  const int nvec = H >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec_total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nvec;
    const int c = (int)(i - row * nvec);
    f32x4 v = {0, 0, 0, 0};
    if (row % S == 0) v = *(const f32x4*)(dcls + (row / S) * H + 4 * c);
    Vec4<T>::store(dh + row * H + 4 * c, v);
  }
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    Vec8<float>::load(src + 8 * i, v);
    Vec8<bf16>::store(dst + 8 * i, v);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n8 << 3; i < n; ++i) dst[i] = (bf16)src[i];
}

// ---- sparse exchange of word-embedding gradient rows (data parallel, large vocabularies; trainer.GradReducer) ----------------
// gather: slot i < n takes row rows[i] of the table (and its id), slots n .. cap-1 are padding (id -1, zeros)
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ rows, int64_t n,
                                                          int64_t* __restrict__ ids_out, float* __restrict__ vals_out, int H) {
  const int64_t i = blockIdx.x;
  const int64_t id = (i < n) ? rows[i] : -1;
  if (threadIdx.x == 0) ids_out[i] = id;
  const int nvec = H >> 2;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    const f32x4 v = (id >= 0) ? *(const f32x4*)(table + id * H + 4 * c) : f32x4{0, 0, 0, 0};
    *(f32x4*)(vals_out + i * H + 4 * c) = v;
  }
}
// table[ids[i]] = 0 (vals == nullptr) or table[ids[i]] += vals[i]; ids unique within one call (negative ids are padding)
__global__ __launch_bounds__(256) void rows_update_kernel(float* __restrict__ table, const int64_t* __restrict__ ids,
                                                          const float* __restrict__ vals, int H) {
  const int64_t i = blockIdx.x;
  const int64_t id = ids[i];
  if (id < 0) return;
  const int nvec = H >> 2;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    float* p = table + id * H + 4 * c;
    *(f32x4*)p = vals ? *(const f32x4*)p + *(const f32x4*)(vals + i * H + 4 * c) : f32x4{0, 0, 0, 0};
  }
}

static inline int vpl_for(int H) { return ((H >> 2) + 63) / 64; }
static inline int grid_rows(int64_t M, int wpb) {
  int64_t g = (M + wpb - 1) / wpb;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

#define DISPATCH_VPL(H, CALL)                                     \
  do {                                                            \
    const int vpl__ = vpl_for(H);                                 \
    if (vpl__ <= 1) { constexpr int VPL = 1; CALL; }              \
    else if (vpl__ <= 2) { constexpr int VPL = 2; CALL; }         \
    else if (vpl__ <= 3) { constexpr int VPL = 3; CALL; }         \
    else if (vpl__ <= 4) { constexpr int VPL = 4; CALL; }         \
    else { constexpr int VPL = 8; CALL; }                         \
  } while (0)

static int check_h(int H) {
  NB_CHECK(H > 0 && H % 4 == 0 && H <= 2048, NBEST_ERR_SHAPE, "hidden size %d must be a multiple of 4 and <= 2048", H);
  return NBEST_OK;
}

// ---- deferred finalizes ----------------------------------------------------------------------------------------------
// A backward layer has four producers of partial rows (LayerNorm-backward x 2, the fused column sums of the dU GEMM, the
// attention backward's bias gradient); finalised one by one that is four 5-us launches of pure latency per layer (49 per
// step).  nbest_encoder_backward opens a batch per layer: while it is open finalize() only RECORDS its job (the producers
// write their partial rows into regions of their own) and one launch at the end of the layer sums them all.
constexpr int kMaxBatch = 8;
struct RowredJob {
  const float* part;
  int nblk, N, nout;
  RowredOut o;
};
struct RowredJobs {
  RowredJob j[kMaxBatch];
};
thread_local RowredJobs* t_batch = nullptr;
thread_local int t_batch_n = 0;

__global__ __launch_bounds__(1024) void rowred_finalize_multi_kernel(RowredJobs jobs) {
  __shared__ float sm[32][33];
  const RowredJob& jb = jobs.j[blockIdx.z];
  const int k = blockIdx.y, N = jb.N, nblk = jb.nblk;
  if (k >= jb.nout || (int)blockIdx.x * 32 >= N) return;        // block-uniform
  float* outp = jb.o.out[k];
  const int col = blockIdx.x * 32 + threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < N && outp != nullptr) {
    const float* p = jb.part + (int64_t)k * nblk * N + col;
    int b = threadIdx.y;
    for (; b + 96 < nblk; b += 128) {
      s0 += p[(int64_t)b * N]; s1 += p[(int64_t)(b + 32) * N]; s2 += p[(int64_t)(b + 64) * N]; s3 += p[(int64_t)(b + 96) * N];
    }
    for (; b < nblk; b += 32) s0 += p[(int64_t)b * N];
  }
  sm[threadIdx.y][threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.y == 0 && col < N && outp != nullptr) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += sm[r][threadIdx.x];
    outp[col] = jb.o.accumulate[k] ? outp[col] + s : s;
  }
}

static int finalize(const float* part, int nblk, int N, float* o0, int a0, float* o1, int a1, float* o2, int a2,
                    hipStream_t st) {
  RowredOut o;
  o.out[0] = o0; o.out[1] = o1; o.out[2] = o2;
  o.accumulate[0] = a0; o.accumulate[1] = a1; o.accumulate[2] = a2;
  const int nout = o2 ? 3 : (o1 ? 2 : 1);
  if (t_batch != nullptr && t_batch_n < kMaxBatch) {              // deferred: summed by nbest_internal_rowred_batch_flush
    RowredJob& jb = t_batch->j[t_batch_n++];
    jb.part = part; jb.nblk = nblk; jb.N = N; jb.nout = nout; jb.o = o;
    return NBEST_OK;
  }
  rowred_finalize_kernel<<<dim3((N + 31) / 32, nout), dim3(32, 32), 0, st>>>(part, nblk, N, o);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

}  // namespace

// Deferred finalizes (see RowredJobs): between begin and flush every finalize of THIS thread is recorded instead of launched; the
// caller guarantees that the recorded producers wrote their partial rows to distinct memory.  flush launches one kernel for all.
static thread_local RowredJobs t_jobs_storage;
void nbest_internal_rowred_batch_begin() { t_batch = &t_jobs_storage; t_batch_n = 0; }
void nbest_internal_rowred_batch_abort() { t_batch = nullptr; t_batch_n = 0; }     // error paths: never leave a batch open
int nbest_internal_rowred_batch_flush(hipStream_t st) {
  RowredJobs* b = t_batch;
  const int n = t_batch_n;
  t_batch = nullptr; t_batch_n = 0;
  if (b == nullptr || n == 0) return NBEST_OK;
  int maxN = 0, maxout = 1;
  for (int i = 0; i < n; ++i) { if (b->j[i].N > maxN) maxN = b->j[i].N; if (b->j[i].nout > maxout) maxout = b->j[i].nout; }
  rowred_finalize_multi_kernel<<<dim3((maxN + 31) / 32, maxout, n), dim3(32, 32), 0, st>>>(*b);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

// out[n] (+)= sum over `nrows` partial rows (used by the fused bias-gradient epilogues of the GEMM / attention)
int nbest_internal_partial_rows_sum(const float* part, int nrows, int N, float* out, int accumulate, hipStream_t st) {
  return finalize(part, nrows, N, out, accumulate, nullptr, 0, nullptr, 0, st);
}

extern "C" size_t nbest_rowred_ws_bytes(int64_t M, int64_t N) {
  int64_t b = (M + 31) / 32;
  if (b < 1) b = 1;
  if (b > kMaxLnBwdBlocks) b = kMaxLnBwdBlocks;
  return (size_t)3 * b * N * sizeof(float);
}
constexpr int kEmbTpc = NBEST_EMB_TPC;   // sorted tokens per wave of embed_bwd_word_kernel
extern "C" size_t nbest_embed_bwd_ws_bytes(int64_t M, int64_t H) {
  // partial rows of embed_bwd_pos_kernel (<= kMaxLnBwdBlocks blocks x 5 quantities) | two partial rows per chunk of embed_bwd_word_kernel
  // | the fix-up work list (a counter + one chunk id per entry)
  const int64_t chunks = (M + kEmbTpc - 1) / kEmbTpc;
  return (size_t)5 * kMaxLnBwdBlocks * H * sizeof(float) + (size_t)chunks * 2 * H * sizeof(float) + (size_t)(chunks + 4) * sizeof(int);
}

// y8 != NULL (bf16, H % 256 == 0, H <= 1024 only): also write the e4m3 copy of y that the next fp8 forward GEMM reads
int nbest_internal_layernorm_fwd8(const void* x, const float* gamma, const float* beta, void* y, void* y8, float* stats,
                                  int64_t M, int H, float eps, int dtype, nbest_stream_t stream, const uint32_t* a_prev, uint32_t* a_new) {
  if (int e = check_h(H)) return e;
  NB_CHECK(x && gamma && beta && y && stats && M > 0, NBEST_ERR_ARG, "layernorm_fwd: null pointer or M <= 0");
  NB_CHECK(!y8 || (dtype == NBEST_BF16 && H % 256 == 0 && H <= 1024), NBEST_ERR_SHAPE, "layernorm_fwd: fp8 copy needs bf16 and H in {256..1024}");
  hipStream_t st = (hipStream_t)stream;
  const int grid0 = grid_rows(M, 4);
  if (dtype == NBEST_F32) {
    DISPATCH_VPL(H, (ln_fwd_kernel<float, VPL><<<grid0, 256, 0, st>>>((const float*)x, gamma, beta, (float*)y, stats, M, H, eps)));
  } else if (dtype == NBEST_BF16 && H % 256 == 0 && H <= 1024) {
    const int grid = grid0 < 1024 ? grid0 : 1024;   // 512 ... 4096 blocks tie at 17 us for 100 MB (5.9 TB/s); 8192: 19.5
    switch (H / 256) {
      case 1: ln_fwd_fast_kernel<1><<<grid, 256, 0, st>>>((const bf16*)x, gamma, beta, (bf16*)y, stats, M, eps, (uint8_t*)y8, a_prev, a_new); break;
      case 2: ln_fwd_fast_kernel<2><<<grid, 256, 0, st>>>((const bf16*)x, gamma, beta, (bf16*)y, stats, M, eps, (uint8_t*)y8, a_prev, a_new); break;
      case 3: ln_fwd_fast_kernel<3><<<grid, 256, 0, st>>>((const bf16*)x, gamma, beta, (bf16*)y, stats, M, eps, (uint8_t*)y8, a_prev, a_new); break;
      default: ln_fwd_fast_kernel<4><<<grid, 256, 0, st>>>((const bf16*)x, gamma, beta, (bf16*)y, stats, M, eps, (uint8_t*)y8, a_prev, a_new); break;
    }
  } else if (dtype == NBEST_BF16) {
    DISPATCH_VPL(H, (ln_fwd_kernel<bf16, VPL><<<grid0, 256, 0, st>>>((const bf16*)x, gamma, beta, (bf16*)y, stats, M, H, eps)));
  } else NB_CHECK(false, NBEST_ERR_DTYPE, "layernorm_fwd: bad dtype %d", dtype);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                   int64_t M, int H, float eps, int dtype, nbest_stream_t stream) {
  return nbest_internal_layernorm_fwd8(x, gamma, beta, y, nullptr, stats, M, H, eps, dtype, stream, nullptr, nullptr);
}

// f8 (bf16 fast path only): e4m3 copy + amax of the gradient the dense-layer GEMMs read (dx_drop under dropout, else dx)
int nbest_internal_layernorm_bwd8(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                                  void* dx_drop, float* dgamma, float* dbeta, float* dbias, int64_t M, int H, int dtype,
                                  int accumulate, float drop_p, uint64_t seed, uint32_t drop_stream, void* ws,
                                  size_t ws_bytes, nbest_stream_t stream, Fp8Grad f8) {
  if (int e = check_h(H)) return e;
  NB_CHECK(!f8.amax_new || (dtype == NBEST_BF16 && H % 256 == 0 && H <= 1024), NBEST_ERR_SHAPE, "layernorm_bwd: fp8 copy needs bf16 and H in {256..1024}");
  NB_CHECK(dy && x && stats && gamma && dx && dgamma && dbeta && ws && M > 0, NBEST_ERR_ARG, "layernorm_bwd: null pointer");
  NB_CHECK(ws_bytes >= nbest_rowred_ws_bytes(M, H), NBEST_ERR_WORKSPACE, "layernorm_bwd: workspace too small");
  const DropCfg d = make_drop(drop_p, seed, drop_stream);
  NB_CHECK(d.thr16 == 0 || (dx_drop && dx_drop != dx) || (!dx_drop && f8.out8), NBEST_ERR_ARG, "layernorm_bwd: dropout needs a separate dx_drop buffer");
  NB_CHECK(M * (int64_t)H < (int64_t)1 << 32 || d.thr16 == 0, NBEST_ERR_SHAPE, "layernorm_bwd: dropout counter overflow");
  hipStream_t st = (hipStream_t)stream;
  int nblk = (int)((M + 31) / 32 < 1 ? 1 : ((M + 31) / 32 > kMaxLnBwdBlocks ? kMaxLnBwdBlocks : (M + 31) / 32));
  if (nblk > 512) nblk = 512;   // isolated, 256 and 512 blocks tie (32 us for 200 MB); inside the step 512 is faster (35 vs 44 us)
  int rpb = (int)((M + nblk - 1) / nblk);
  nblk = (int)((M + rpb - 1) / rpb);
#ifndef NBEST_LN_WAVES
#define NBEST_LN_WAVES 4
#endif
  constexpr int waves = NBEST_LN_WAVES;   // 8: measured in the step, see profiles/README.md
  float* part = (float*)ws;
  const size_t smem = (size_t)3 * H * sizeof(float);
  const int wb = dbias ? 1 : 0;
  if (dtype == NBEST_F32) {
    DISPATCH_VPL(H, (ln_bwd_kernel<float, VPL><<<nblk, 256, smem, st>>>((const float*)dy, (const float*)x, stats, gamma,
                                                                         (float*)dx, (float*)dx_drop, part, M, H, rpb, wb, d)));
  } else if (dtype == NBEST_BF16 && H % 256 == 0 && H <= 1024) {
#define NB_LNB(V, D, B)                                                                                                        \
  do {                                                                                                                         \
    if (waves == 8) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)ln_bwd_fast_kernel<V, D, B, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 3 * H * 4); \
      ln_bwd_fast_kernel<V, D, B, 8><<<nblk, 512, 8 * 3 * H * 4, st>>>((const bf16*)dy, (const bf16*)x, stats, gamma, (bf16*)dx,       \
                                                                       (bf16*)dx_drop, part, M, rpb, d, f8);                   \
    } else {                                                                                                                   \
      ln_bwd_fast_kernel<V, D, B, 4><<<nblk, 256, 4 * 3 * H * 4, st>>>((const bf16*)dy, (const bf16*)x, stats, gamma, (bf16*)dx,       \
                                                                       (bf16*)dx_drop, part, M, rpb, d, f8);                   \
    }                                                                                                                          \
  } while (0)
#define NB_LNB_V(V)                                                     \
  do {                                                                  \
    if (d.thr16) { if (wb) NB_LNB(V, true, true); else NB_LNB(V, true, false); } \
    else { if (wb) NB_LNB(V, false, true); else NB_LNB(V, false, false); }       \
  } while (0)
    switch (H / 256) {
      case 1: NB_LNB_V(1); break;
      case 2: NB_LNB_V(2); break;
      case 3: NB_LNB_V(3); break;
      default: NB_LNB_V(4); break;
    }
#undef NB_LNB_V
#undef NB_LNB
  } else if (dtype == NBEST_BF16) {
    DISPATCH_VPL(H, (ln_bwd_kernel<bf16, VPL><<<nblk, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, stats, gamma,
                                                                        (bf16*)dx, (bf16*)dx_drop, part, M, H, rpb, wb, d)));
  } else NB_CHECK(false, NBEST_ERR_DTYPE, "layernorm_bwd: bad dtype %d", dtype);
  NB_LAUNCH_CHECK();
  return finalize(part, nblk, H, dgamma, accumulate, dbeta, accumulate, dbias, accumulate, st);
}

extern "C" int nbest_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                                   void* dx_drop, float* dgamma, float* dbeta, float* dbias, int64_t M, int H, int dtype,
                                   int accumulate, float drop_p, uint64_t seed, uint32_t drop_stream, void* ws,
                                   size_t ws_bytes, nbest_stream_t stream) {
  return nbest_internal_layernorm_bwd8(dy, x, stats, gamma, dx, dx_drop, dgamma, dbeta, dbias, M, H, dtype, accumulate, drop_p, seed,
                                       drop_stream, ws, ws_bytes, stream, Fp8Grad{nullptr, nullptr, nullptr});
}

extern "C" int nbest_colsum(const void* X, float* out, int64_t M, int64_t N, int64_t ld, int dtype, int accumulate,
                            void* ws, size_t ws_bytes, nbest_stream_t stream) {
  NB_CHECK(X && out && ws && M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, NBEST_ERR_ARG, "colsum: bad arguments");
  NB_CHECK(ws_bytes >= nbest_rowred_ws_bytes(M, N), NBEST_ERR_WORKSPACE, "colsum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = rowred_blocks(M);
  const int rpb = (int)((M + nblk - 1) / nblk);
  float* part = (float*)ws;
  dim3 grid((unsigned)((N + 255) / 256), nblk);
  if (dtype == NBEST_F32) colsum_kernel<float><<<grid, 256, 0, st>>>((const float*)X, part, M, N, ld, rpb);
  else if (dtype == NBEST_BF16) colsum_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)X, part, M, N, ld, rpb);
  else NB_CHECK(false, NBEST_ERR_DTYPE, "colsum: bad dtype %d", dtype);
  NB_LAUNCH_CHECK();
  return finalize(part, nblk, (int)N, out, accumulate, nullptr, 0, nullptr, 0, st);
}

extern "C" int nbest_embed_ln_fwd(const int64_t* ids, const int64_t* seg, const int64_t* pos, const void* word,
                                  const void* type, const void* ptab, const float* gamma, const float* beta, void* out,
                                  float* stats, int64_t M, int H, float eps, int dtype, float drop_p, uint64_t seed,
                                  uint32_t drop_stream, nbest_stream_t stream) {
  if (int e = check_h(H)) return e;
  NB_CHECK(ids && pos && word && type && ptab && gamma && beta && out && stats && M > 0, NBEST_ERR_ARG, "embed_ln_fwd: null pointer");
  NB_CHECK(M * (int64_t)H < (int64_t)1 << 32, NBEST_ERR_SHAPE, "embed_ln_fwd: M*H must fit 32 bits (dropout counter)");
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop_p, seed, drop_stream);
  const int grid = grid_rows(M, 4);
  if (dtype == NBEST_F32) {
    DISPATCH_VPL(H, (embed_fwd_kernel<float, VPL><<<grid, 256, 0, st>>>(ids, seg, pos, (const float*)word, (const float*)type,
                                                                        (const float*)ptab, gamma, beta, (float*)out, stats, M, H, eps, d)));
  } else if (dtype == NBEST_BF16) {
    DISPATCH_VPL(H, (embed_fwd_kernel<bf16, VPL><<<grid, 256, 0, st>>>(ids, seg, pos, (const bf16*)word, (const bf16*)type,
                                                                       (const bf16*)ptab, gamma, beta, (bf16*)out, stats, M, H, eps, d)));
  } else NB_CHECK(false, NBEST_ERR_DTYPE, "embed_ln_fwd: bad dtype %d", dtype);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_embed_ln_bwd(const int64_t* ids, const int64_t* seg, const int64_t* pos, const int32_t* perm, const void* word,
                                  const void* type, const void* ptab, const float* gamma, const float* stats,
                                  const void* dout, float* dword, float* dtype_tab, float* dptab, float* dgamma,
                                  float* dbeta, int B, int S, int H, int n_types, int dtype, int64_t word_pad_id,
                                  int64_t pos_pad_id, int accumulate, int tables_accumulate, float drop_p, uint64_t seed,
                                  uint32_t drop_stream, void* ws, size_t ws_bytes, nbest_stream_t stream) {
  if (int e = check_h(H)) return e;
  const int64_t M = (int64_t)B * S;
  NB_CHECK(ids && pos && word && type && ptab && gamma && stats && dout && dword && dtype_tab && dptab && dgamma && dbeta && ws && M > 0,
           NBEST_ERR_ARG, "embed_ln_bwd: null pointer");
  NB_CHECK(perm, NBEST_ERR_ARG, "embed_ln_bwd: perm (stable argsort of ids, int32 [B*S]) is required: the word-table gradient is a "
                                "segmented reduce over the sorted tokens, not a scatter of atomics");
  NB_CHECK(M < ((int64_t)1 << 31), NBEST_ERR_SHAPE, "embed_ln_bwd: B*S must fit int32 (perm)");
  NB_CHECK(ws_bytes >= nbest_embed_bwd_ws_bytes(M, H), NBEST_ERR_WORKSPACE, "embed_ln_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop_p, seed, drop_stream);
  // S x Y blocks (position, batch slice): as many as the partial-row workspace holds (kMaxLnBwdBlocks), at most one per sample
  int Y = kMaxLnBwdBlocks / S;
  if (Y > B) Y = B;
  if (Y < 1) Y = 1;
  NB_CHECK((int64_t)S * Y <= kMaxLnBwdBlocks, NBEST_ERR_SHAPE, "embed_ln_bwd: S = %d exceeds %d", S, kMaxLnBwdBlocks);
  float* part = (float*)ws;
  float* wpart = part + (size_t)5 * kMaxLnBwdBlocks * H;
  const size_t smem = (size_t)5 * H * sizeof(float);
  const dim3 grid(S, Y);
  const int64_t chunks = (M + kEmbTpc - 1) / kEmbTpc;
  const unsigned wgrid = (unsigned)((chunks + 3) / 4);
  int* fixlist = (int*)(wpart + (size_t)chunks * 2 * H);
#define NB_EMB_BWD2(TT, FULL)                                                                                                   \
  DISPATCH_VPL(H, ((void)hipFuncSetAttribute((const void*)embed_bwd_pos_kernel<TT, VPL, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), \
                   embed_bwd_pos_kernel<TT, VPL, FULL><<<grid, 256, smem, st>>>(ids, seg, pos, (const TT*)word, (const TT*)type, (const TT*)ptab, gamma, \
                                                                                 stats, (const TT*)dout, dtype_tab, dptab, part, fixlist, B, S, H, pos_pad_id, d), \
                   embed_bwd_word_kernel<TT, VPL, FULL><<<wgrid, 256, 0, st>>>(perm, ids, seg, pos, (const TT*)word, (const TT*)type, (const TT*)ptab, \
                                                                                gamma, stats, (const TT*)dout, dword, wpart, fixlist, M, H,  \
                                                                                kEmbTpc, word_pad_id, tables_accumulate, d)))
#define NB_EMB_BWD(TT)                                   \
  do {                                                   \
    if (H % 256 == 0 && H <= 1024) NB_EMB_BWD2(TT, true); \
    else NB_EMB_BWD2(TT, false);                         \
  } while (0)
  if (dtype == NBEST_F32) NB_EMB_BWD(float);
  else if (dtype == NBEST_BF16) NB_EMB_BWD(bf16);
  else NB_CHECK(false, NBEST_ERR_DTYPE, "embed_ln_bwd: bad dtype %d", dtype);
#undef NB_EMB_BWD2
#undef NB_EMB_BWD
  NB_LAUNCH_CHECK();
  embed_bwd_wordfix_kernel<<<(unsigned)(chunks < 2048 ? chunks : 2048), 256, (size_t)4 * H * sizeof(float), st>>>(perm, ids, wpart, fixlist, dword, M, H,
                                                                                                                    kEmbTpc, tables_accumulate);
  embed_bwd_finalize_kernel<<<dim3((H + 31) / 32, 4 + (S + 31) / 32), dim3(32, 32), 0, st>>>(part, pos, S, Y, H, n_types, pos_pad_id, dgamma, dbeta,
                                                                                             dtype_tab, dptab, accumulate, tables_accumulate);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_rows_gather(const float* table, const int64_t* rows, int64_t n_rows, int64_t cap, int64_t* ids_out,
                                 float* vals_out, int H, nbest_stream_t stream) {
  NB_CHECK(table && ids_out && vals_out && (rows || n_rows == 0) && n_rows >= 0 && cap >= n_rows && H > 0 && H % 4 == 0, NBEST_ERR_ARG,
           "rows_gather: bad arguments");
  if (cap == 0) return NBEST_OK;
  rows_gather_kernel<<<(unsigned)cap, 256, 0, (hipStream_t)stream>>>(table, rows, n_rows, ids_out, vals_out, H);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_rows_zero(float* table, const int64_t* ids, int64_t n, int H, nbest_stream_t stream) {
  NB_CHECK(table && (ids || n == 0) && n >= 0 && H > 0 && H % 4 == 0, NBEST_ERR_ARG, "rows_zero: bad arguments");
  if (n == 0) return NBEST_OK;
  rows_update_kernel<<<(unsigned)n, 256, 0, (hipStream_t)stream>>>(table, ids, nullptr, H);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_rows_add(float* table, const int64_t* ids, const float* vals, int64_t n, int H, nbest_stream_t stream) {
  NB_CHECK(table && ((ids && vals) || n == 0) && n >= 0 && H > 0 && H % 4 == 0, NBEST_ERR_ARG, "rows_add: bad arguments");
  if (n == 0) return NBEST_OK;
  rows_update_kernel<<<(unsigned)n, 256, 0, (hipStream_t)stream>>>(table, ids, vals, H);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_cls_grad_scatter(const float* dcls, void* dhidden, int B, int S, int H, int dtype,
                                      nbest_stream_t stream) {
  if (int e = check_h(H)) return e;
  NB_CHECK(dcls && dhidden && B > 0 && S > 0, NBEST_ERR_ARG, "cls_grad_scatter: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nv = (int64_t)B * S * (H >> 2);
  int grid = (int)((nv + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (dtype == NBEST_F32) cls_scatter_kernel<float><<<grid, 256, 0, st>>>(dcls, (float*)dhidden, nv, S, H);
  else if (dtype == NBEST_BF16) cls_scatter_kernel<bf16><<<grid, 256, 0, st>>>(dcls, (bf16*)dhidden, nv, S, H);
  else NB_CHECK(false, NBEST_ERR_DTYPE, "cls_grad_scatter: bad dtype %d", dtype);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_cast_f32_to_bf16(const float* src, void* dst, int64_t n, nbest_stream_t stream) {
  NB_CHECK(src && dst && n >= 0, NBEST_ERR_ARG, "cast: bad arguments");
  if (n == 0) return NBEST_OK;
  int64_t g = ((n >> 3) + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  cast_bf16_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(src, (bf16*)dst, n);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}
