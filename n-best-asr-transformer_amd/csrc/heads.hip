// K7 STC heads + losses (forward and analytic backward), K8 CLS-MSE, K10 decode.
//
// Shapes are tiny (B x 171 x 768 for bert-base on DSTC2), so everything is fp32 VALU work in a
// handful of launches; the point is to replace ~40 tiny launches + 4 host syncs of the reference
// (/root/reference/n_best_asr_bert.py:160-195) and to hand dCLS to the encoder backward.
//
// Workspace layout (floats): cls[B][H] | logits[B][R] | dz[B][R] | sample_loss[B][4]
#include "common.h"

namespace {

__device__ __forceinline__ int layer_of_row(int r, const int32_t* __restrict__ head_row, int n_top, int* lay_cache) {
  (void)lay_cache;
  if (r < n_top) return 0;
  // heads are laid out in increasing top order: find the top t with head_row[t] <= r < next head's start
  int lay = 0, k = 0;
  for (int t = 0; t < n_top; ++t) {
    const int hr = head_row[t];
    if (hr >= 0) {
      ++k;
      if (hr <= r) lay = k;
    }
  }
  return lay;
}

// The 1 + n_heads dropout masks of a step (one per linear layer of the hierarchical classifier, each over [B][H]) are hashed ONCE
// and kept as bits: mw[(layer * B + b) * W + (h >> 5)], W = ceil(H / 32) words per row.  Every head row of a layer shares its
// layer's mask, so hashing per (row, b, h) in the three kernels below repeated the same 14-instruction decision R / n_layers = 14
// times (34 M hashes per kernel at B = 256); the bits are the same decisions (nb_keep), so nothing changes numerically.
__device__ __forceinline__ void heads_mask_words(const DropCfg& drop, int lay, int b, int B, int H, int W, uint32_t* lds_row,
                                                 uint32_t* __restrict__ glob_row) {
  const int lane = threadIdx.x & 63;
  for (int h0 = (threadIdx.x & ~63); h0 < H; h0 += blockDim.x) {
    const int h = h0 + lane;
    const bool keep = (h < H) && nb_keep(drop, (uint32_t)((lay * B + b) * H + h));
    const uint64_t bal = __ballot(keep);
    if (lane == 0) {
      const int w = h0 >> 5;
      lds_row[w] = (uint32_t)bal;
      if (w + 1 < W) lds_row[w + 1] = (uint32_t)(bal >> 32);
      if (glob_row) {
        glob_row[w] = (uint32_t)bal;
        if (w + 1 < W) glob_row[w + 1] = (uint32_t)(bal >> 32);
      }
    }
  }
}

// logits[b][r] = sum_h Wh[r][h] * drop_{layer(r)}(cls[b][h]) + bh[r];  grid (B, kLogitSplit), block 256:
// blockIdx.y owns a contiguous slice of the head rows (more, shorter blocks: the kernel is latency bound).  A block hashes the
// masks of the layers its rows belong to into LDS; the blockIdx.y == 0 block of a sample hashes all of them and also writes
// them to `mw` for the two gradient kernels.
constexpr int kLogitSplit = 8;
// (Measured: 1024-thread blocks make the three GEMV-like kernels of the heads slower - 67 / 33 / 48 us against 46 / 39 / 29.)
constexpr int kHeadsThreads = 256;
template <typename T>
__global__ __launch_bounds__(kHeadsThreads) void heads_logits_kernel(const T* __restrict__ hidden, int64_t cls_stride,
                                                           const float* __restrict__ Wh, const float* __restrict__ bh,
                                                           const int32_t* __restrict__ head_row, int n_top, int R, int H,
                                                           float* __restrict__ cls_out, float* __restrict__ logits,
                                                           uint32_t* __restrict__ mw, int32_t* __restrict__ lay_row, int n_lay,
                                                           DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [H] | mask words [n_lay][W] | layer of row [R]
  const int b = blockIdx.x, B = gridDim.x, W = (H + 31) >> 5;
  uint32_t* mk = (uint32_t*)(xs + H);
  int32_t* lay_s = (int32_t*)(mk + n_lay * W);
  const T* x = hidden + (int64_t)b * cls_stride;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    const float v = to_f<T>(x[h]);
    xs[h] = v;
    if (blockIdx.y == 0) cls_out[(int64_t)b * H + h] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int per = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = (r0 + per < R) ? r0 + per : R;
  // layer_of_row walks head_row with dependent loads (microseconds): once per row, all rows in parallel, then LDS lookups; the
  // first block also publishes the table for the gradient kernels
  const bool first = (blockIdx.x == 0 && blockIdx.y == 0);
  for (int r = (first ? 0 : r0) + (int)threadIdx.x; r < (first ? R : r1); r += blockDim.x) {
    const int l = layer_of_row(r, head_row, n_top, nullptr);
    lay_s[r] = l;
    if (first) lay_row[r] = l;
  }
  __syncthreads();
  if (drop.thr16) {
    const bool all = (blockIdx.y == 0);
    const int la = all ? 0 : (r0 < r1 ? lay_s[r0] : 0);
    const int lb = all ? n_lay - 1 : (r0 < r1 ? lay_s[r1 - 1] : -1);
    for (int lay = la; lay <= lb; ++lay)
      heads_mask_words(drop, lay, b, B, H, W, mk + lay * W, all ? mw + ((int64_t)lay * B + b) * W : nullptr);
  }
  __syncthreads();
  for (int r = r0 + wave; r < r1; r += nw) {
    const int lay = lay_s[r];
    const float* w = Wh + (int64_t)r * H;
    const uint32_t* mrow = mk + lay * W;
    float s = 0.f;
#pragma unroll 16
    for (int h = lane; h < H; h += 64) {
      float xv = xs[h];
      if (drop.thr16) xv = ((mrow[h >> 5] >> (h & 31)) & 1u) ? xv * drop.scale : 0.f;
      s = fmaf(w[h], xv, s);
    }
    s = wave_sum(s);
    if (lane == 0) logits[(int64_t)b * R + r] = s + bh[r];
  }
}

// per sample: scores, losses and d(loss)/d(logits).  grid B, block 64 (one wave; n_top, head sizes small)
__global__ __launch_bounds__(64) void heads_scores_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                          const int32_t* __restrict__ bottom_off, const int32_t* __restrict__ bottom_ids,
                                                          const int32_t* __restrict__ head_row, int n_top, int n_bottom, int R,
                                                          int n_heads, float* __restrict__ top, float* __restrict__ bott,
                                                          float* __restrict__ fin, float* __restrict__ dz,
                                                          float* __restrict__ sample_loss) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* z = logits + (int64_t)b * R;
  const float* y = labels + (int64_t)b * n_bottom;
  float l_bot = 0.f, l_top = 0.f, l_ce = 0.f;
  // top labels are processed one per iteration by the whole wave (lanes parallelise the head columns)
  for (int t = 0; t < n_top; ++t) {
    const int o0 = bottom_off[t], nk = bottom_off[t + 1] - o0, hr = head_row[t];
    const float zt = z[t];
    const float pt = 1.0f / (1.0f + __expf(-zt));
    float dpt = 0.f;  // d(loss)/d(top score)
    float ytop = 0.f;
    if (hr < 0) {
      // single bottom label: final = top score
      const int bi = bottom_ids[o0];
      const float yy = y[bi];
      ytop = yy;
      if (lane == 0) {
        fin[(int64_t)b * n_bottom + bi] = pt;
        l_bot += -(yy * fmaxf(logf(pt), -100.f) + (1.f - yy) * fmaxf(logf(1.f - pt), -100.f));
      }
      dpt += (pt - yy) / fmaxf(pt * (1.f - pt), 1e-12f);
    } else {
      // softmax head over nk columns (nk may exceed 64: strided)
      float mx = -INFINITY;
      for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, z[hr + j]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int j = lane; j < nk; j += 64) se += __expf(z[hr + j] - mx);
      se = wave_sum(se);
      const float inv = 1.0f / se;
      // class index: position of the active bottom label, else the last column (NONE)
      int idx = nk - 1;
      float ysum = 0.f;
      for (int j = lane; j < nk; j += 64) {
        const float yy = y[bottom_ids[o0 + j]];
        ysum += yy;
        if (yy > 0.5f) idx = min(idx, j);   // STC_util asserts at most one hot: first == the one
      }
      ysum = wave_sum(ysum);
      for (int o = 32; o > 0; o >>= 1) idx = min(idx, __shfl_xor(idx, o, 64));
      if (ysum == 0.f) idx = nk - 1;
      ytop = ysum;
      // pass 1: ds_j and sum_j ds_j s_j ; bottom BCE ; gradient wrt top through final = pt * s_j
      float dot = 0.f, dtop_acc = 0.f, lb = 0.f;
      for (int j = lane; j < nk; j += 64) {
        const float s = __expf(z[hr + j] - mx) * inv;
        const int bi = bottom_ids[o0 + j];
        const float yy = y[bi];
        const float f = pt * s;
        fin[(int64_t)b * n_bottom + bi] = f;
        bott[(int64_t)b * (R - n_top) + (hr - n_top) + j] = s;
        lb += -(yy * fmaxf(logf(f), -100.f) + (1.f - yy) * fmaxf(logf(1.f - f), -100.f));
        const float gf = (f - yy) / fmaxf(f * (1.f - f), 1e-12f);
        float ds = gf * pt;
        if (j == idx) ds += -1.0f / ((s + 1e-12f) * (float)n_heads);
        dot += ds * s;
        dtop_acc += gf * s;
      }
      dot = wave_sum(dot);
      dpt += wave_sum(dtop_acc);
      l_bot += wave_sum(lb) * (lane == 0 ? 1.f : 0.f);
      for (int j = lane; j < nk; j += 64) {
        const float s = __expf(z[hr + j] - mx) * inv;
        const int bi = bottom_ids[o0 + j];
        const float yy = y[bi];
        const float f = pt * s;
        const float gf = (f - yy) / fmaxf(f * (1.f - f), 1e-12f);
        float ds = gf * pt;
        if (j == idx) {
          ds += -1.0f / ((s + 1e-12f) * (float)n_heads);
          l_ce += -logf(s + 1e-12f) / (float)n_heads;
        }
        dz[(int64_t)b * R + hr + j] = s * (ds - dot);
      }
    }
    // top BCE against y . B2T
    if (lane == 0) l_top += -(ytop * fmaxf(logf(pt), -100.f) + (1.f - ytop) * fmaxf(logf(1.f - pt), -100.f));
    dpt += (pt - ytop) / fmaxf(pt * (1.f - pt), 1e-12f);
    if (lane == 0) {
      top[(int64_t)b * n_top + t] = pt;
      dz[(int64_t)b * R + t] = dpt * pt * (1.f - pt);
    }
  }
  l_ce = wave_sum(l_ce);
  if (lane == 0) {
    sample_loss[4 * b + 0] = l_bot;
    sample_loss[4 * b + 1] = l_top;
    sample_loss[4 * b + 2] = l_ce;
    sample_loss[4 * b + 3] = 0.f;
  }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ sample_loss, int B, float* __restrict__ out) {
  __shared__ float sm[16];
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) s += sample_loss[4 * b + k];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) out[k] = s;
  }
}

// dWh[r][h] = sum_b dz[b][r] * drop(cls[b][h]); dbh[r] = sum_b dz[b][r].  grid (R, ceil(H/64)), block 256:
// a block owns one head row x 64 columns (lane = column), its waves split the batch.  Fixed-order LDS reduce (deterministic).
__global__ __launch_bounds__(kHeadsThreads) void heads_wgrad_kernel(const float* __restrict__ cls, const float* __restrict__ dz,
                                                          const int32_t* __restrict__ head_row, int n_top, int B, int R, int H,
                                                          float* __restrict__ dWh, float* __restrict__ dbh, int accumulate,
                                                          const uint32_t* __restrict__ mw, const int32_t* __restrict__ lay_row,
                                                          DropCfg drop) {
  __shared__ float red[kHeadsThreads / 64][64];
  __shared__ float redb[kHeadsThreads / 64];
  const int r = blockIdx.x, h = blockIdx.y * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int lay = lay_row[r];
  const int W = (H + 31) >> 5;
  const uint32_t* mcol = mw + (int64_t)lay * B * W + (h >> 5);   // the layer's mask bits of column h: one word per sample
  float acc = 0.f, sb = 0.f;
  if (h < H) {
#pragma unroll 16
    for (int b = wave; b < B; b += nw) {
      const float g = dz[(int64_t)b * R + r];
      float xv = cls[(int64_t)b * H + h];
      if (drop.thr16) xv = ((mcol[(int64_t)b * W] >> (h & 31)) & 1u) ? xv * drop.scale : 0.f;
      acc = fmaf(g, xv, acc);
      sb += g;
    }
  }
  red[wave][threadIdx.x & 63] = acc;
  if ((threadIdx.x & 63) == 0) redb[wave] = sb;
  __syncthreads();
  if (wave == 0 && h < H) {
    const int l = threadIdx.x;
    float v = 0.f;
    for (int w = 0; w < nw; ++w) v += red[w][l];
    float* o = dWh + (int64_t)r * H + h;
    *o = accumulate ? *o + v : v;
  }
  if (threadIdx.x == 0 && blockIdx.y == 0) {
    float v = 0.f;
    for (int w = 0; w < nw; ++w) v += redb[w];
    dbh[r] = accumulate ? dbh[r] + v : v;
  }
}

// dcls[b][h] = sum_r dz[b][r] * Wh[r][h] * dropmask_{layer(r)}(b,h).  grid (B, ceil(H/64)), block 256: waves split r
__global__ __launch_bounds__(kHeadsThreads) void heads_dgrad_kernel(const float* __restrict__ Wh, const float* __restrict__ dz,
                                                          const int32_t* __restrict__ head_row, int n_top, int R, int H,
                                                          float* __restrict__ dcls, const uint32_t* __restrict__ mw,
                                                          const int32_t* __restrict__ lay_row, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // dz row [R] | layer of row [R] | red [waves][64]
  const int b = blockIdx.x, B = gridDim.x;
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.y * 64 + l, nw = blockDim.x >> 6;
  float* red = sh + 2 * R;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    sh[r] = dz[(int64_t)b * R + r];
    sh[R + r] = (float)lay_row[r];
  }
  __syncthreads();
  float s = 0.f;
  const int W = (H + 31) >> 5;
  const uint32_t* mcol = mw + (int64_t)b * W + (h >> 5);       // this sample's mask bits of column h: one word per layer
  if (h < H) {
#pragma unroll 16
    for (int r = wave; r < R; r += nw) {
      float w = Wh[(int64_t)r * H + h];
      if (drop.thr16) w = ((mcol[(int64_t)(int)sh[R + r] * B * W] >> (h & 31)) & 1u) ? w * drop.scale : 0.f;
      s = fmaf(sh[r], w, s);
    }
  }
  red[wave * 64 + l] = s;
  __syncthreads();
  if (wave == 0 && h < H) {
    float v = 0.f;
    for (int w = 0; w < nw; ++w) v += red[w * 64 + l];
    dcls[(int64_t)b * H + h] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cls_mse_kernel(const T* __restrict__ ha, int64_t sa, const T* __restrict__ ht, int64_t st_,
                                                      float* __restrict__ loss, float* __restrict__ da, float* __restrict__ dt,
                                                      int B, int H, float grad_scale) {
  __shared__ float sm[16];
  const int64_t n = (int64_t)B * H;
  const float c = 2.0f * grad_scale / (float)n;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const int b = (int)(i / H), h = (int)(i - (int64_t)b * H);
    const float d = to_f<T>(ha[b * sa + h]) - to_f<T>(ht[b * st_ + h]);
    s += d * d;
    if (da) da[i] += c * d;
    if (dt) dt[i] = -c * d;
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) loss[0] = s / (float)n;
}

__global__ void decode_kernel(const float* __restrict__ top, const float* __restrict__ bott, const int32_t* __restrict__ bottom_off,
                              const int32_t* __restrict__ bottom_ids, const int32_t* __restrict__ head_row,
                              const uint8_t* __restrict__ none_flag, int n_top, int R, int32_t* __restrict__ pred, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n_top) return;
  const int b = i / n_top, t = i - b * n_top;
  int out = -1;
  if (top[i] > 0.5f) {
    const int o0 = bottom_off[t], nk = bottom_off[t + 1] - o0, hr = head_row[t];
    if (hr < 0) out = bottom_ids[o0];
    else {
      const float* s = bott + (int64_t)b * (R - n_top) + (hr - n_top);
      int am = 0;
      float best = s[0];
      for (int j = 1; j < nk; ++j)
        if (s[j] > best) { best = s[j]; am = j; }   // first maximum, as numpy argmax
      const int bi = bottom_ids[o0 + am];
      out = none_flag[bi] ? -1 : bi;
    }
  }
  pred[i] = out;
}

}  // namespace

extern "C" size_t nbest_heads_ws_bytes(int B, int R, int H) {
  // cls | logits | dz | per-sample losses | dropout mask bits of at most R + 1 classifier layers
  return ((size_t)B * H + (size_t)2 * B * R + (size_t)4 * B) * sizeof(float) + (size_t)(R + 1) * B * ((H + 31) / 32) * sizeof(uint32_t) + (size_t)R * sizeof(int32_t);
}

extern "C" int nbest_stc_heads(const void* hidden, int64_t cls_stride, const float* Wh, const float* bh,
                               const nbest_label_space* ls, const float* labels, float* top, float* bott,
                               float* final_scores, float* loss_parts, float* dcls, float* dWh, float* dbh, int B, int H,
                               int dtype, int need_grad, int accumulate, float drop_p, uint64_t seed, uint32_t drop_stream,
                               void* ws, size_t ws_bytes, nbest_stream_t stream) {
  NB_CHECK(hidden && Wh && bh && ls && labels && top && bott && final_scores && loss_parts && ws && B > 0 && H > 0,
           NBEST_ERR_ARG, "stc_heads: null pointer");
  NB_CHECK(!need_grad || (dcls && dWh && dbh), NBEST_ERR_ARG, "stc_heads: need_grad without gradient buffers");
  const int R = ls->n_rows, n_top = ls->n_top, n_bottom = ls->n_bottom;
  NB_CHECK(R > n_top && n_top > 0 && n_bottom > 0 && R <= 4096 && H <= 2048, NBEST_ERR_SHAPE, "stc_heads: bad label space");
  NB_CHECK(ws_bytes >= nbest_heads_ws_bytes(B, R, H), NBEST_ERR_WORKSPACE, "stc_heads: workspace too small");
  NB_CHECK((int64_t)11 * B * H < ((int64_t)1 << 32), NBEST_ERR_SHAPE, "stc_heads: B*H too large");
  hipStream_t st = (hipStream_t)stream;
  float* cls = (float*)ws;
  float* logits = cls + (size_t)B * H;
  float* dz = logits + (size_t)B * R;
  float* sloss = dz + (size_t)B * R;
  uint32_t* mw = (uint32_t*)(sloss + (size_t)4 * B);
  int32_t* lay_row = (int32_t*)(mw + (size_t)(R + 1) * B * ((H + 31) / 32));
  const DropCfg d = make_drop(drop_p, seed, drop_stream);
  // number of softmax heads = R - n_top rows split over heads; the CE term averages over heads: count on host is not
  // available (device arrays), so it is passed implicitly: n_heads = #tops with head_row >= 0, computed by the caller
  // into ls->n_bottom? -> no: derive it from sizes: sum over multi tops (nk) = R - n_top and singles = n_top - n_heads,
  // n_bottom = (R - n_top) + (n_top - n_heads)  =>  n_heads = R - n_bottom.
  const int n_heads = R - n_bottom;
  NB_CHECK(n_heads > 0, NBEST_ERR_SHAPE, "stc_heads: label space has no multi-value head");
  const int n_lay = n_heads + 1;
  const size_t smemH = (size_t)H * sizeof(float) + (size_t)n_lay * ((H + 31) / 32) * sizeof(uint32_t) + (size_t)R * sizeof(int32_t);
  if (dtype == NBEST_F32)
    heads_logits_kernel<float><<<dim3(B, kLogitSplit), kHeadsThreads, smemH, st>>>((const float*)hidden, cls_stride, Wh, bh, ls->head_row, n_top, R, H, cls, logits, mw, lay_row, n_lay, d);
  else if (dtype == NBEST_BF16)
    heads_logits_kernel<bf16><<<dim3(B, kLogitSplit), kHeadsThreads, smemH, st>>>((const bf16*)hidden, cls_stride, Wh, bh, ls->head_row, n_top, R, H, cls, logits, mw, lay_row, n_lay, d);
  else NB_CHECK(false, NBEST_ERR_DTYPE, "stc_heads: bad dtype %d", dtype);
  NB_LAUNCH_CHECK();
  heads_scores_kernel<<<B, 64, 0, st>>>(logits, labels, ls->bottom_off, ls->bottom_ids, ls->head_row, n_top, n_bottom, R,
                                       n_heads, top, bott, final_scores, dz, sloss);
  NB_LAUNCH_CHECK();
  loss_reduce_kernel<<<1, 256, 0, st>>>(sloss, B, loss_parts);
  NB_LAUNCH_CHECK();
  if (need_grad) {
    heads_wgrad_kernel<<<dim3(R, (H + 63) / 64), kHeadsThreads, 0, st>>>(cls, dz, ls->head_row, n_top, B, R, H, dWh, dbh, accumulate, mw, lay_row, d);
    NB_LAUNCH_CHECK();
    heads_dgrad_kernel<<<dim3(B, (H + 63) / 64), kHeadsThreads, ((size_t)2 * R + kHeadsThreads) * sizeof(float), st>>>(Wh, dz, ls->head_row, n_top, R, H, dcls, mw, lay_row, d);
    NB_LAUNCH_CHECK();
  }
  return NBEST_OK;
}

extern "C" int nbest_cls_mse(const void* hidden_a, int64_t stride_a, const void* hidden_t, int64_t stride_t, float* loss,
                             float* da, float* dt, int B, int H, int dtype, float grad_scale, nbest_stream_t stream) {
  NB_CHECK(hidden_a && hidden_t && loss && B > 0 && H > 0, NBEST_ERR_ARG, "cls_mse: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == NBEST_F32)
    cls_mse_kernel<float><<<1, 256, 0, st>>>((const float*)hidden_a, stride_a, (const float*)hidden_t, stride_t, loss, da, dt, B, H, grad_scale);
  else if (dtype == NBEST_BF16)
    cls_mse_kernel<bf16><<<1, 256, 0, st>>>((const bf16*)hidden_a, stride_a, (const bf16*)hidden_t, stride_t, loss, da, dt, B, H, grad_scale);
  else NB_CHECK(false, NBEST_ERR_DTYPE, "cls_mse: bad dtype %d", dtype);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

extern "C" int nbest_stc_decode(const float* top, const float* bott, const nbest_label_space* ls, const uint8_t* none_flag,
                                int32_t* pred, int B, nbest_stream_t stream) {
  NB_CHECK(top && bott && ls && none_flag && pred && B > 0, NBEST_ERR_ARG, "stc_decode: null pointer");
  const int n = B * ls->n_top;
  decode_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(top, bott, ls->bottom_off, ls->bottom_ids, ls->head_row,
                                                                 none_flag, ls->n_top, ls->n_rows, pred, B);
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}
