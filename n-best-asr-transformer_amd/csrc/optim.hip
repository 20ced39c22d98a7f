// K9 multi-tensor BertAdam over flat fp32 arenas (p, g, m, v) + bf16 compute-copy refresh.
// Restates /root/reference/models/optimization.py:237-302 under the per-parameter grouping of
// /root/reference/n_best_asr_bert.py:540-561 (each tensor clipped on its own to L2 norm 1.0).
//
// HBM-bound: phase 1 reads g (4 B/param) for the per-tensor norms, phase 3 reads p,g,m,v and writes
// p,m,v (+2 B bf16 copy) = 30 B/param.  All tensors are processed by ONE launch per phase: the block
// -> (tensor, chunk) map is a binary search over desc.block_start (tensors ordered by offset).
#include "common.h"

namespace {

constexpr int kChunk = 16384;  // elements per block: 256 threads x 16 float4

__device__ __forceinline__ int find_tensor(const nbest_tensor_desc* __restrict__ d, int n, int blk) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].block_start <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, const nbest_tensor_desc* __restrict__ descs,
                                                    int n_tensors, float* __restrict__ partial, int blk_off) {
  __shared__ float sm[16];
  const int blk = blockIdx.x + blk_off;          // a rank of a sharded optimizer runs only its own range of blocks
  const int t = find_tensor(descs, n_tensors, blk);
  const nbest_tensor_desc d = descs[t];
  float s = 0.f;
  if (d.active) {
    const int64_t c0 = (int64_t)(blk - d.block_start) * kChunk;
    const int64_t c1 = (c0 + kChunk < d.numel) ? c0 + kChunk : d.numel;
    const float* gp = g + d.offset;
    const bool vec = ((d.offset & 3) == 0);
    if (vec) {
      const int64_t v1 = c0 + ((c1 - c0) & ~(int64_t)3);
      for (int64_t i = c0 + 4 * threadIdx.x; i < v1; i += 1024) {
        f32x4 x = *(const f32x4*)(gp + i);
        s += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
      }
      for (int64_t i = v1 + threadIdx.x; i < c1; i += 256) s += gp[i] * gp[i];
    } else {
      for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) s += gp[i] * gp[i];
    }
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) partial[blk] = s;
}

// one wave per tensor: coef[t] = min(1, max_norm / (||g_t|| + 1e-6))
__global__ __launch_bounds__(64) void clip_coef_kernel(const float* __restrict__ partial, const nbest_tensor_desc* __restrict__ descs,
                                                       int n_tensors, int n_blocks, float max_norm, float* __restrict__ coef) {
  const int t = blockIdx.x;
  const int b0 = descs[t].block_start;
  const int b1 = (t + 1 < n_tensors) ? descs[t + 1].block_start : n_blocks;
  float s = 0.f;
  for (int b = b0 + threadIdx.x; b < b1; b += 64) s += partial[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    float c = 1.f;
    if (max_norm > 0.f) c = fminf(max_norm / (sqrtf(s) + 1e-6f), 1.0f);
    coef[t] = c;
  }
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float coef, float b1, float b2, float eps,
                                      float lr, float wd) {
  g *= coef;
  m = m * b1 + (1.f - b1) * g;
  v = v * b2 + (1.f - b2) * g * g;
  float u = m / (sqrtf(v) + eps);
  if (wd > 0.f) u += wd * p;
  p -= lr * u;
}

__global__ __launch_bounds__(256) void bertadam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, bf16* __restrict__ plow,
                                                       const nbest_tensor_desc* __restrict__ descs, int n_tensors,
                                                       const float* __restrict__ coef, float lr_mult, float b1, float b2, float eps,
                                                       int blk_off) {
  const int blk = blockIdx.x + blk_off;
  const int t = find_tensor(descs, n_tensors, blk);
  const nbest_tensor_desc d = descs[t];
  if (!d.active) return;
  const float cf = coef[t];
  const float lr = d.lr * lr_mult;
  const int64_t c0 = (int64_t)(blk - d.block_start) * kChunk;
  const int64_t c1 = (c0 + kChunk < d.numel) ? c0 + kChunk : d.numel;
  const int64_t base = d.offset;
  const bool vec = ((base & 3) == 0);
  int64_t v1 = c0;
  if (vec) {
    v1 = c0 + ((c1 - c0) & ~(int64_t)3);
    for (int64_t i = c0 + 4 * threadIdx.x; i < v1; i += 1024) {
      f32x4 pp = *(f32x4*)(p + base + i), gg = *(const f32x4*)(g + base + i);
      f32x4 mm = *(f32x4*)(m + base + i), vv = *(f32x4*)(v + base + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float p1 = pp[e], m1 = mm[e], v1e = vv[e];
        adam1(p1, gg[e], m1, v1e, cf, b1, b2, eps, lr, d.wd);
        pp[e] = p1; mm[e] = m1; vv[e] = v1e;
      }
      *(f32x4*)(p + base + i) = pp;
      *(f32x4*)(m + base + i) = mm;
      *(f32x4*)(v + base + i) = vv;
      if (plow) Vec4<bf16>::store(plow + base + i, pp);
    }
  }
  for (int64_t i = v1 + threadIdx.x; i < c1; i += 256) {
    float pp = p[base + i], mm = m[base + i], vv = v[base + i];
    adam1(pp, g[base + i], mm, vv, cf, b1, b2, eps, lr, d.wd);
    p[base + i] = pp; m[base + i] = mm; v[base + i] = vv;
    if (plow) plow[base + i] = (bf16)pp;
  }
}

// bf16 [rows][cols] -> [cols][rows] for a table of matrices living at the same element offsets in two
// arenas (the k-contiguous weight copy the dgrad GEMMs read).  One launch: block -> (matrix, 64x64 tile).
__global__ __launch_bounds__(256) void transpose_multi_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                              const nbest_matrix_desc* __restrict__ descs, int n) {
  __shared__ __attribute__((aligned(16))) bf16 tile[64][72];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const nbest_matrix_desc d = descs[lo];
  const int t = blockIdx.x - d.tile_start, tc = (d.cols + 63) / 64;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const bf16* s = src + d.offset;
  bf16* o = dst + d.offset;
  if (r0 + 64 <= d.rows && c0 + 64 <= d.cols && ((d.rows | d.cols | d.offset) & 7) == 0) {   // 16-byte path (every encoder matrix)
    transpose_tile64<bf16>(s, o, d.rows, d.cols, r0, c0, &tile[0][0]);
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < d.rows && c < d.cols) ? s[(int64_t)r * d.cols + c] : (bf16)0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < d.cols && r < d.rows) o[(int64_t)c * d.rows + r] = tile[tx][ty + 4 * i];
  }
}

}  // namespace

extern "C" int nbest_transpose_weights(const void* src, void* dst, const nbest_matrix_desc* descs, int n_matrices, int n_tiles,
                                       nbest_stream_t stream) {
  NB_CHECK(src && dst && descs && n_matrices > 0 && n_tiles > 0, NBEST_ERR_ARG, "transpose_weights: bad arguments");
  transpose_multi_kernel<<<n_tiles, 256, 0, (hipStream_t)stream>>>((const bf16*)src, (bf16*)dst, descs, n_matrices);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_bertadam_chunk(void) { return kChunk; }

// The step in its two halves, each over a RANGE of blocks [blk_lo, blk_hi) - the sharded optimizer of the data-parallel path
// (nbest_amd/optim.py): a rank computes the block sums of squares of its own blocks (`partial` [n_blocks], zero elsewhere; the host
// SUM-all-reduces it: x + 0 is exact, so every rank ends up with the very numbers a single process computes), then the clip
// coefficients of ALL tensors (identical on every rank) and the update of its own blocks.
extern "C" int nbest_bertadam_norms(const float* g, const nbest_tensor_desc* descs, int n_tensors, int n_blocks, int blk_lo, int blk_hi,
                                    float* partial, nbest_stream_t stream) {
  NB_CHECK(g && descs && partial && n_tensors > 0 && 0 <= blk_lo && blk_lo <= blk_hi && blk_hi <= n_blocks, NBEST_ERR_ARG,
           "bertadam_norms: bad arguments");
  if (blk_hi == blk_lo) return NBEST_OK;
  sumsq_kernel<<<blk_hi - blk_lo, 256, 0, (hipStream_t)stream>>>(g, descs, n_tensors, partial, blk_lo);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_bertadam_update(float* p, const float* g, float* m, float* v, void* p_lowp, const nbest_tensor_desc* descs,
                                     int n_tensors, int n_blocks, int blk_lo, int blk_hi, const float* partial, float* coef,
                                     float lr_mult, float b1, float b2, float eps, float max_grad_norm, nbest_stream_t stream) {
  NB_CHECK(p && g && m && v && descs && partial && coef && n_tensors > 0 && 0 <= blk_lo && blk_lo <= blk_hi && blk_hi <= n_blocks,
           NBEST_ERR_ARG, "bertadam_update: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  clip_coef_kernel<<<n_tensors, 64, 0, st>>>(partial, descs, n_tensors, n_blocks, max_grad_norm, coef);
  NB_LAUNCH_CHECK();
  if (blk_hi == blk_lo) return NBEST_OK;
  bertadam_kernel<<<blk_hi - blk_lo, 256, 0, st>>>(p, g, m, v, (bf16*)p_lowp, descs, n_tensors, coef, lr_mult, b1, b2, eps, blk_lo);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_bertadam_step(float* p, float* g, float* m, float* v, void* p_lowp, const nbest_tensor_desc* descs,
                                   int n_tensors, int n_blocks, float lr_mult, float b1, float b2, float eps, float max_grad_norm,
                                   void* ws, size_t ws_bytes, nbest_stream_t stream) {
  NB_CHECK(p && g && m && v && descs && ws && n_tensors > 0 && n_blocks > 0, NBEST_ERR_ARG, "bertadam: null pointer");
  NB_CHECK(ws_bytes >= ((size_t)n_blocks + n_tensors) * sizeof(float), NBEST_ERR_WORKSPACE, "bertadam: workspace too small");
  float* partial = (float*)ws;
  float* coef = partial + n_blocks;
  if (int rc = nbest_bertadam_norms(g, descs, n_tensors, n_blocks, 0, n_blocks, partial, stream)) return rc;
  return nbest_bertadam_update(p, g, m, v, p_lowp, descs, n_tensors, n_blocks, 0, n_blocks, partial, coef, lr_mult, b1, b2, eps,
                               max_grad_norm, stream);
}
