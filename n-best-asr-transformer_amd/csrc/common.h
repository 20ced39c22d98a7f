// Shared device/host helpers for libnbest_hip.so (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/nbest_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

#define LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))

// ---- error plumbing (host) -------------------------------------------------------------------
void nbest_set_error(const char* fmt, ...);
#define NB_CHECK(cond, code, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      nbest_set_error(__VA_ARGS__);          \
      return (code);                         \
    }                                        \
  } while (0)
#define NB_LAUNCH_CHECK()                                                       \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      nbest_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,            \
                      hipGetErrorString(e__));                                  \
      return NBEST_ERR_LAUNCH;                                                  \
    }                                                                           \
  } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- scalar conversions ----------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// 4-element vector load/store of T as float4 (16 B for fp32, 8 B for bf16)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  typedef f32x4 raw_t;
  static __device__ __forceinline__ raw_t raw_load(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ f32x4 cvt(raw_t v) { return v; }
  static __device__ __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Vec4<bf16> {
  typedef bf16x4 raw_t;
  static __device__ __forceinline__ raw_t raw_load(const bf16* p) { return *(const bf16x4*)p; }
  static __device__ __forceinline__ f32x4 cvt(raw_t v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
  static __device__ __forceinline__ f32x4 load(const bf16* p) {
    bf16x4 v = *(const bf16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  static __device__ __forceinline__ void store(bf16* p, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *(bf16x4*)p = o;
  }
};
// 8-element vector load/store of T as 2 x float4
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float* o) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};
template <> struct Vec8<bf16> {
  static __device__ __forceinline__ void load(const bf16* p, float* o) {
    bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  static __device__ __forceinline__ void store(bf16* p, const float* v) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
    *(bf16x8*)p = o;
  }
};

// cache-policy bits of the LDS-DMA operand loads (aux of raw_ptr_buffer_load_lds: 1 = sc0, 2 = nt, 16 = sc1); experiments only
#ifndef NBEST_DIAG
#define NB_AUX_A 0
#define NB_AUX_B 0
#else
#define NB_AUX_A ((NBEST_DIAG & 8192) ? 2 : 0)
#define NB_AUX_B ((NBEST_DIAG & 16384) ? 2 : 0)
#endif

// Tile id -> (tile_m, tile_n) with the tile columns blocked in groups of `gn`: every tile row of one column group is visited
// before the next group, so the group's slice of the WEIGHT operand ([gn * BN][K]) stays in the 4 MiB L2 of the XCD while the
// activation panel streams past it.  Row-major order (gn = tiles_n) re-reads the whole weight matrix once per tile row, and
// a 3072 x 768 bf16 matrix (4.7 MB) does not fit the L2: the PMC counters showed 140-230 MB of such re-reads per GEMM.
__device__ __forceinline__ void nb_tile_coords(int t, int tiles_m, int gn, int& tm, int& tn) {
  const int per_group = tiles_m * gn;
  const int g = t / per_group, r = t - g * per_group;
  tm = r / gn;
  tn = g * gn + (r - tm * gn);
}
// host: the largest divisor of tiles_n whose weight slice (cols_per_tile * K * elem_bytes each) stays under `limit_kb`.  Measured
// on the step (same-box A/B): bf16 - halves of the 4.7 MB FFN matrices and thirds of the 3.5 MB QKV matrix pay (limit 2400),
// finer groups do not (every extra group re-reads the activations); fp8 - 1.2 MB slices (limit 1600) beat the whole 2.4 MB matrix.
inline int nb_group_cols(int64_t tiles_n, int64_t slice_bytes_per_tile, int64_t limit_kb) {
  int best = 1;
  for (int d = 1; d <= tiles_n; ++d)
    if (tiles_n % d == 0 && d * slice_bytes_per_tile <= limit_kb * 1024) best = d;
#ifdef NBEST_EXPERIMENTS
  if (const char* e = getenv("NBEST_GN")) { const int v = atoi(e); if (v > 0 && tiles_n % v == 0) best = v; else if (v == 0) best = (int)tiles_n; }
#endif
  return best;
}

// Streaming (nontemporal) store: for GEMM / attention outputs of tens to hundreds of MB, which otherwise wash the operands
// other tiles still read out of the 4 MiB L2 of every XCD.  The shipped library streams EVERY GEMM output (nb_stream_output), so
// the hint is unconditional there: `__builtin_nontemporal_store` (global_store ... nt), which the compiler schedules and whose
// data registers it tracks.  (Round 2 selected it at run time through an inline-asm store inside an if / else: every store became
// its own basic block with an `s_waitcnt vmcnt(0)` at the join - harmless behind an LDS restage that serialised the epilogue
// anyway, 50 us per launch once the epilogue became one block of independent work.)  Experiment builds keep the run-time switch.
template <typename V> __device__ __forceinline__ void st_stream(V* p, V v, bool nt) {
  static_assert(sizeof(V) == 8 || sizeof(V) == 16, "st_stream: 8- or 16-byte vectors");
#ifdef NBEST_EXPERIMENTS
  if (!nt) { *p = v; return; }
  if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
#else
  (void)nt;
  __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ void st_stream_bf16x8(bf16* p, const float* v, bool nt) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
  st_stream((bf16x8*)p, o, nt);
}
// Range-checked, nontemporal buffer stores / loads for the register epilogues of the bf16 GEMMs: the descriptor covers the M valid
// rows, the hardware drops stores (returns zero for loads) past its end - ragged row tiles cost no branch, and without branches
// the whole epilogue is one basic block whose loads, arithmetic and stores the compiler interleaves freely.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
template <int AUX = 2 /* nt */>
__device__ __forceinline__ void nb_bstore_bf16x8(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, const float* v) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, byte_off, 0, AUX);
}
__device__ __forceinline__ void nb_bstore8(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, uint32_t lo, uint32_t hi) {
  __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo, hi}, rs, byte_off, 0, 2 /* nt */);
}

// host side: which GEMM outputs are streamed (bytes of the bf16 output)
inline bool nb_stream_output(int64_t out_bytes) {
  int64_t min_mb = 0;
#ifdef NBEST_EXPERIMENTS
  if (const char* e = getenv("NBEST_NT_MIN_MB")) min_mb = atoll(e);
#endif
  return out_bytes >= (min_mb << 20);
}

// ---- 64x64 tile transpose of a row-major matrix with 1- or 2-byte elements (transposed weight copies) ----------------------
// 256 threads; 16-byte global loads and stores, the transposition happens in the LDS image (element-wise writes into
// tile[col][row], 16-byte reads of a transposed row).  For a tile fully inside a matrix whose dimensions and base are multiples
// of 16 elements; the callers keep an element-wise path for everything else.  `tile`: 64 x (64 + 16 / sizeof(T)) elements.
template <typename T>
__device__ __forceinline__ void transpose_tile64(const T* __restrict__ s, T* __restrict__ o, int rows, int cols, int r0, int c0,
                                                 T* tile) {
  constexpr int EPV = 16 / (int)sizeof(T);         // elements per 16-byte vector: 16 (e4m3) or 8 (bf16)
  constexpr int PITCH = 64 + EPV;
  constexpr int CPR = 64 / EPV;                    // vectors per tile row
  typedef __attribute__((ext_vector_type(EPV))) T vec_t;
#pragma unroll
  for (int it = 0; it < (64 * CPR) / 256; ++it) {
    const int idx = it * 256 + threadIdx.x, r = idx / CPR, ch = idx % CPR;
    const vec_t v = *(const vec_t*)(s + (int64_t)(r0 + r) * cols + c0 + ch * EPV);
#pragma unroll
    for (int e = 0; e < EPV; ++e) tile[(ch * EPV + e) * PITCH + r] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < (64 * CPR) / 256; ++it) {
    const int idx = it * 256 + threadIdx.x, c = idx / CPR, ch = idx % CPR;
    *(vec_t*)(o + (int64_t)(c0 + c) * rows + r0 + ch * EPV) = *(const vec_t*)(tile + c * PITCH + ch * EPV);
  }
}

// ---- wave64 / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// two full-wave sums at once: four DPP steps inside each row of 16 lanes (no LDS crossbar), then two xor-shuffles
// across the rows; the two dependency chains interleave.  Every lane gets both totals.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  a = dpp_add<0xB1>(a); b = dpp_add<0xB1>(b);     // quad_perm [1,0,3,2]
  a = dpp_add<0x4E>(a); b = dpp_add<0x4E>(b);     // quad_perm [2,3,0,1]
  a = dpp_add<0x141>(a); b = dpp_add<0x141>(b);   // row_half_mirror
  a = dpp_add<0x140>(a); b = dpp_add<0x140>(b);   // row_mirror
  a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
  a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
}
__device__ __forceinline__ float wave_sum_dpp(float a) {
  a = dpp_add<0xB1>(a); a = dpp_add<0x4E>(a); a = dpp_add<0x141>(a); a = dpp_add<0x140>(a);
  a += __shfl_xor(a, 16, 64);
  a += __shfl_xor(a, 32, 64);
  return a;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); smem needs >= 16 floats. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* smem) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}

// ---- GELU (erf form, as the reference's encoder: hidden_act="gelu") ----------------------------
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float u) {
  // d/du [u * Phi(u)] = Phi(u) + u * phi(u)
  const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
  return cdf + u * pdf;
}

// Fast forms for the bf16 MFMA epilogues (VALU-bound otherwise: erff costs ~40 instructions).
// erf by Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7 - two orders below bf16 resolution; GELU and
// GELU' share the single exp(-u^2/2).  The fp32 parity path keeps erff.
__device__ __forceinline__ void gelu_parts_fast(float u, float& cdf, float& e) {
  const float z = fabsf(u) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  e = __expf(-z * z);                               // exp(-u^2/2)
  const float erf_abs = fmaf(-poly * t, e, 1.0f);   // erf(|u|/sqrt2)
  cdf = fmaf(0.5f, copysignf(erf_abs, u), 0.5f);
}
// The same on a PAIR of values with packed fp32 math (v_pk_mul / v_pk_fma: two lanes' worth per instruction); returns
// gelu(u) in h and gelu'(u) in g.  The GEMM epilogues that apply GELU are VALU-bound: this form is what they call.
__device__ __forceinline__ void gelu_pair_fast(f32x2 u, f32x2& h, f32x2& g) {
  const f32x2 au = {fabsf(u[0]), fabsf(u[1])};
  const f32x2 d = au * 0.23164190f + 1.0f;                               // 1 + 0.3275911 |u| / sqrt2
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 poly = t * (-0.5f * 1.061405429f) + (0.5f * 1.453152027f);       // the polynomial times -1/2 (folded into its coefficients)
  poly = poly * t + (-0.5f * 1.421413741f);
  poly = poly * t + (0.5f * 0.284496736f);
  poly = poly * t + (-0.5f * 0.254829592f);
  const f32x2 x2 = (u * u) * (-0.5f * 1.4426950408889634f);              // exp(-u^2/2) = exp2(-u^2/2 log2 e)
  const f32x2 e = {__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])};
  const f32x2 half_erf = (poly * t) * e + 0.5f;                          // erf(|u|/sqrt2) / 2
  const f32x2 cdf = {0.5f + copysignf(half_erf[0], u[0]), 0.5f + copysignf(half_erf[1], u[1])};
  g = (u * 0.39894228040143267794f) * e + cdf;
  h = u * cdf;
}
__device__ __forceinline__ float gelu_fast(float u) {
  float cdf, e;
  gelu_parts_fast(u, cdf, e);
  return u * cdf;
}
__device__ __forceinline__ float dgelu_fast(float u) {
  float cdf, e;
  gelu_parts_fast(u, cdf, e);
  return fmaf(u * 0.39894228040143267794f, e, cdf);
}

// GELU'(u), kept for the backward, travels in 8 BITS in the bf16 path: fixed point q = round(200 g') + 26, i.e. a step of
// 1/200 over [-0.13, 1.145] (g' lies in [-0.129, 1.129]) with 0 and 1 represented EXACTLY (saturated units and dead units
// carry no error); |error| <= 0.0025, the size of a bf16 ulp at 1.  Halves the bytes of the largest tensor the forward writes
// only for the backward to read once.
// v_cvt_pk_u8_f32 converts (round to nearest even, saturating to [0, 255]: tools/micro/cvt_u8_probe.hip) AND inserts the byte:
// one instruction per value where clamp + convert + shift/or took four.
__device__ __forceinline__ uint32_t gd_pack4(const float* g) {
  uint32_t w = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[e], 200.f, 26.f), e, w);
  return w;
}
__device__ __forceinline__ void gd_unpack4(uint32_t w, float* g) {
#pragma unroll
  for (int e = 0; e < 4; ++e) g[e] = fmaf((float)((w >> (8 * e)) & 255u), 0.005f, -0.13f);
}

// four floats -> four OCP e4m3 bytes (unit scale, saturating at +-448): operands of the fp8 forward GEMMs
__device__ __forceinline__ uint32_t fp8_pack4(const float* v) {
  float c[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(v[e], -448.f, 448.f);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
  return (uint32_t)w;
}

// fp8 copy of a GRADIENT tensor for the fp8 dgrad GEMMs: e4m3(g * s) with the per-tensor power-of-two scale
// s = 2^floor(log2(56 / amax)) taken from the amax the SAME tensor had in the previous backward pass (delayed scaling);
// the producer also records this pass's amax (bits of a non-negative float order like the float).  All-null = off.
// Headroom: the history is ONE step old, so the scale maps it to 56 = 448 / 8 - a gradient may grow 8 x from one step to the
// next (first real batch after the calibration pass, a loss spike) before anything saturates; the price is 2 of the 17.8 bits
// of e4m3 range at the bottom (elements below amax * 2^-14.8 flush), where a gradient tensor's mass is not (round 2 kept 2 x
// headroom, 224, and saturated silently; tests/test_model_gpu.py::test_fp8w_gradient_amax_jump).
// Weights (static, re-quantised from the master after every step) keep 2 x: fp8_scale_of.
struct Fp8Grad {
  uint8_t* out8;
  const uint32_t* amax_prev;
  uint32_t* amax_new;
};
// 2^floor(log2(target / amax)), exponent clamped to +-100; a zero, denormal (< 2^-100) or non-finite amax gives scale 1
__host__ __device__ __forceinline__ float fp8_pow2_scale(float amax, float target) {
  if (!(amax >= 7.8886e-31f) || !(amax <= 3.0e38f)) return 1.f;      // also catches NaN
  const float e = floorf(log2f(target / amax));
  return exp2f(e < -100.f ? -100.f : (e > 100.f ? 100.f : e));
}
__host__ __device__ __forceinline__ float fp8_scale_of(float amax) { return fp8_pow2_scale(amax, 224.f); }        // weights
__host__ __device__ __forceinline__ float fp8_gscale_of(float amax) { return fp8_pow2_scale(amax, 56.f); }        // gradients
__device__ __forceinline__ float fp8_grad_scale(const uint32_t* amax_prev) { return amax_prev ? fp8_gscale_of(__uint_as_float(*amax_prev)) : 1.f; }
// Forward ACTIVATIONS of the fp8 mode (x, ctx, x1, gelu(u)): the same delayed per-tensor scale, but with 2 x headroom (the amax maps to
// 112 .. 224, as the weights).  Activations move slowly from one step to the next, and what a scale costs is at the BOTTOM: every
// halving of the scale pushes another octave of the small entries into e4m3's subnormals.  Measured on the outlier-statistics golden
// (amax 260 under a bulk of O(1)): with the gradients' 8 x headroom the scaled copy was WORSE than the unit-scale one it replaces
// (score floor 8.7e-2 against 6.1e-2) - the bulk paid for headroom the outliers did not need.
__host__ __device__ __forceinline__ float fp8_ascale_of(float amax) { return fp8_pow2_scale(amax, 224.f); }
__device__ __forceinline__ float fp8_act_scale(const uint32_t* amax_prev) { return amax_prev ? fp8_ascale_of(__uint_as_float(*amax_prev)) : 1.f; }
// Recording a tensor's amax: *slot = max(*slot, v) for one lane of a wave.  An amax "new" pointer names a SLOT BLOCK of
// NBEST_AMAX_TENSOR_WORDS words (include/nbest_hip.h): kAmaxSlots words 256 bytes apart, one picked by the block index; the
// caller folds them (nbest_fp8_amax_fold) into the single word the next pass reads.  Why: device-scope accesses to ONE word
// are served at about 1 ns each wherever they come from - an unconditional atomicMax per wave cost +66 us on a 126 us kernel
// (12 288 waves), and even with the read-first filter below the 65 536 waves of the e4m3 cast paid 100 us for their reads of
// the one word (12 -> 115 us; LayerNorm forward +10 us, attention forward +13 us).  Sixteen words in different channels
// divide that by sixteen; the read-first filter still lets all but the waves that raise their slot skip the atomic.
constexpr int kAmaxSlots = 16, kAmaxSlotStride = NBEST_AMAX_TENSOR_WORDS / 16;
__device__ __forceinline__ void amax_update(uint32_t* p, float v) {
  p += ((blockIdx.x + 5 * blockIdx.y + 3 * blockIdx.z) & (kAmaxSlots - 1)) * kAmaxSlotStride;
  const uint32_t b = __float_as_uint(v);
  if (b > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, b);
}

// ---- counter-based dropout ---------------------------------------------------------------------
// keep-decision for element `idx` of stream `stream` under (seed): 16-bit uniform compared with a
// 16-bit threshold.  thr16 = round(p * 65536); the effective drop probability is thr16/65536 and
// the survivor scale is 65536/(65536-thr16) (computed on the host, passed as `scale`).
// One 32-bit hash yields the decisions of the element PAIR (idx & ~1, idx | 1).
__device__ __forceinline__ uint32_t nb_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint32_t thr16;   // 0 => dropout disabled
  float scale;      // 1/(1-p_eff)
  uint32_t key;     // hash(seed, stream)
};
__host__ __device__ __forceinline__ uint32_t nb_mix_key(uint64_t seed, uint32_t stream) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(stream + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (uint32_t)(z ^ (z >> 31));
}
__device__ __forceinline__ bool nb_keep(const DropCfg& d, uint32_t idx) {
  const uint32_t h = nb_hash32((idx >> 1) * 0x9E3779B9U + d.key);
  const uint32_t u = (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
  return u >= d.thr16;
}
// decisions for the element pair idx, idx + 1 (idx even): bit i set = keep
__device__ __forceinline__ uint32_t nb_keep2(const DropCfg& d, uint32_t idx) {
  const uint32_t h = nb_hash32((idx >> 1) * 0x9E3779B9U + d.key);
  return ((h & 0xFFFFu) >= d.thr16 ? 1u : 0u) | ((h >> 16) >= d.thr16 ? 2u : 0u);
}
// decisions for 4 consecutive elements idx..idx+3 (idx % 4 == 0): bit i set = keep
__device__ __forceinline__ uint32_t nb_keep4(const DropCfg& d, uint32_t idx) {
  const uint32_t h0 = nb_hash32((idx >> 1) * 0x9E3779B9U + d.key);
  const uint32_t h1 = nb_hash32(((idx >> 1) + 1) * 0x9E3779B9U + d.key);
  return ((h0 & 0xFFFFu) >= d.thr16 ? 1u : 0u) | ((h0 >> 16) >= d.thr16 ? 2u : 0u) |
         ((h1 & 0xFFFFu) >= d.thr16 ? 4u : 0u) | ((h1 >> 16) >= d.thr16 ? 8u : 0u);
}
static inline DropCfg make_drop(float p, uint64_t seed, uint32_t stream) {
  DropCfg d;
  uint32_t thr = (p <= 0.f) ? 0u : (uint32_t)lrintf(p * 65536.0f);
  if (thr > 65535u) thr = 65535u;
  d.thr16 = thr;
  d.scale = 65536.0f / (float)(65536u - thr);
  d.key = nb_mix_key(seed, stream);
  return d;
}
