// K7 STC heads + losses (forward and analytic backward), K8 CLS-MSE, K10 decode.
//
// Shapes are tiny (B x 171 x 768 for bert-base on DSTC2), so everything is fp32 VALU work in two launches
// (heads_fwd_kernel, heads_bwd_kernel); the point is to replace ~40 tiny launches + 4 host syncs of the reference
// (/root/reference/models/modules/hierarchical_classifier.py:35-60, /root/reference/n_best_asr_bert.py:160-195) and to hand
// dCLS to the encoder backward.
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

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ sample_loss, int B, float* __restrict__ out) {
  __shared__ float sm[16];
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) s += sample_loss[4 * b + k];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) out[k] = s;
  }
}

// ---- round 4: the heads as TWO launches (147 us in four launches for a 256 x 768 x 171 problem) ------------------
// heads_fwd_kernel: one block per sample does what heads_logits_kernel + heads_scores_kernel did - CLS row and the layers' dropout bits
// into LDS, the 171 logits (a wave per row), then scores / losses / d(loss)/d(logits) with the TOP labels spread over the waves (the
// old scores kernel walked the 30 tops one after the other in a single wave: 37 us of dependent exp / log / shuffle latency).
// heads_bwd_kernel: weight gradient and input gradient in one grid (blocks [0, nW): an 8-row x 64-column tile of dWh, waves split the
// batch, the CLS column chunk loaded once per sample for all 8 rows; blocks [nW, nW + nD): 8 samples x 64 columns of dCLS, waves split
// the head rows, the weight element loaded once for all 8 samples); the last block sums the per-sample losses.  Fixed-order sums.
constexpr int kFwdThreads = 512;
template <typename T>
__global__ __launch_bounds__(kFwdThreads) void heads_fwd_kernel(const T* __restrict__ hidden, int64_t cls_stride, const float* __restrict__ Wh,
                                                                 const float* __restrict__ bh, const float* __restrict__ labels,
                                                                 const int32_t* __restrict__ bottom_off, const int32_t* __restrict__ bottom_ids,
                                                                 const int32_t* __restrict__ head_row, int n_top, int n_bottom, int R, int H,
                                                                 int n_heads, int n_lay, float* __restrict__ cls_out, float* __restrict__ top,
                                                                 float* __restrict__ bott, float* __restrict__ fin, float* __restrict__ dz,
                                                                 float* __restrict__ sample_loss, uint32_t* __restrict__ mw,
                                                                 int32_t* __restrict__ lay_row, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [H] | logits [R] | layer of row [R] | mask words [n_lay][W] | loss partials [waves][3]
  const int b = blockIdx.x, B = gridDim.x, W = (H + 31) >> 5;
  float* zs = xs + H;
  int32_t* lay_s = (int32_t*)(zs + R);
  uint32_t* mk = (uint32_t*)(lay_s + R);
  float* lsum = (float*)(mk + n_lay * W);
  const T* x = hidden + (int64_t)b * cls_stride;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    const float v = to_f<T>(x[h]);
    xs[h] = v;
    cls_out[(int64_t)b * H + h] = v;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const int l = layer_of_row(r, head_row, n_top, nullptr);
    lay_s[r] = l;
    if (b == 0) lay_row[r] = l;
  }
  if (drop.thr16)
    for (int lay = 0; lay < n_lay; ++lay) heads_mask_words(drop, lay, b, B, H, W, mk + lay * W, mw + ((int64_t)lay * B + b) * W);
  __syncthreads();
  for (int r = wave; r < R; r += nw) {
    const float* w = Wh + (int64_t)r * H;
    const uint32_t* mrow = mk + lay_s[r] * W;
    float s = 0.f;
#pragma unroll 4
    for (int h = lane; h < H; h += 64) {
      float xv = xs[h];
      if (drop.thr16) xv = ((mrow[h >> 5] >> (h & 31)) & 1u) ? xv * drop.scale : 0.f;
      s = fmaf(w[h], xv, s);
    }
    s = wave_sum(s);
    if (lane == 0) zs[r] = s + bh[r];
  }
  __syncthreads();
  const float* z = zs;
  const float* y = labels + (int64_t)b * n_bottom;
  float l_bot = 0.f, l_top = 0.f, l_ce = 0.f;
  for (int t = wave; t < n_top; t += nw) {                  // a wave per top label (lanes parallelise the head columns)
    const int o0 = bottom_off[t], nk = bottom_off[t + 1] - o0, hr = head_row[t];
    const float zt = z[t];
    const float pt = 1.0f / (1.0f + __expf(-zt));
    float dpt = 0.f;  // d(loss)/d(top score)
    float ytop = 0.f;
    if (hr < 0) {
      const int bi = bottom_ids[o0];                        // single bottom label: final = top score
      const float yy = y[bi];
      ytop = yy;
      if (lane == 0) {
        fin[(int64_t)b * n_bottom + bi] = pt;
        l_bot += -(yy * fmaxf(logf(pt), -100.f) + (1.f - yy) * fmaxf(logf(1.f - pt), -100.f));
      }
      dpt += (pt - yy) / fmaxf(pt * (1.f - pt), 1e-12f);
    } else {
      float mx = -INFINITY;                                 // softmax head over nk columns (nk may exceed 64: strided)
      for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, z[hr + j]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int j = lane; j < nk; j += 64) se += __expf(z[hr + j] - mx);
      se = wave_sum(se);
      const float inv = 1.0f / se;
      int idx = nk - 1;                                     // class index: the active bottom label, else the last column (NONE)
      float ysum = 0.f;
      for (int j = lane; j < nk; j += 64) {
        const float yy = y[bottom_ids[o0 + j]];
        ysum += yy;
        if (yy > 0.5f) idx = min(idx, j);
      }
      ysum = wave_sum(ysum);
      for (int o = 32; o > 0; o >>= 1) idx = min(idx, __shfl_xor(idx, o, 64));
      if (ysum == 0.f) idx = nk - 1;
      ytop = ysum;
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
    if (lane == 0) l_top += -(ytop * fmaxf(logf(pt), -100.f) + (1.f - ytop) * fmaxf(logf(1.f - pt), -100.f));   // top BCE against y . B2T
    dpt += (pt - ytop) / fmaxf(pt * (1.f - pt), 1e-12f);
    if (lane == 0) {
      top[(int64_t)b * n_top + t] = pt;
      dz[(int64_t)b * R + t] = dpt * pt * (1.f - pt);
    }
  }
  l_ce = wave_sum(l_ce);
  if (lane == 0) { lsum[3 * wave] = l_bot; lsum[3 * wave + 1] = l_top; lsum[3 * wave + 2] = l_ce; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = 0.f;
    for (int w = 0; w < nw; ++w) v += lsum[3 * w + threadIdx.x];     // wave order: the same sum on every run
    sample_loss[4 * b + threadIdx.x] = v;
  }
  if (threadIdx.x == 3) sample_loss[4 * b + 3] = 0.f;
}

constexpr int kBwdRows = 8;      // head rows (wgrad) / samples (dgrad) per block
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ cls, const float* __restrict__ dz, const float* __restrict__ Wh,
                                                        int B, int R, int H, float* __restrict__ dWh, float* __restrict__ dbh,
                                                        float* __restrict__ dcls, int accumulate, const uint32_t* __restrict__ mw,
                                                        const int32_t* __restrict__ lay_row, DropCfg drop, const float* __restrict__ sample_loss,
                                                        float* __restrict__ loss_out, int nW) {
  __shared__ float red[4][kBwdRows][64];
  __shared__ float redb[4][kBwdRows];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nh = (H + 63) >> 6, W = (H + 31) >> 5;
  const int bid = blockIdx.x;
  if (bid < nW) {
    // ---- dWh[r][h] = sum_b dz[b][r] * drop_{layer(r)}(cls[b][h]); dbh[r] = sum_b dz[b][r] ----
    const int r0 = (bid / nh) * kBwdRows, h = (bid % nh) * 64 + lane;
    const bool hv = h < H;
    int lay[kBwdRows];
#pragma unroll
    for (int j = 0; j < kBwdRows; ++j) lay[j] = lay_row[min(r0 + j, R - 1)];
    float acc[kBwdRows], sb[kBwdRows];
#pragma unroll
    for (int j = 0; j < kBwdRows; ++j) acc[j] = sb[j] = 0.f;
    for (int b = wave; b < B; b += 4) {
      const float xv = hv ? cls[(int64_t)b * H + h] : 0.f;
      const float xd = xv * drop.scale;
#pragma unroll
      for (int j = 0; j < kBwdRows; ++j) {
        const float g = (r0 + j < R) ? dz[(int64_t)b * R + r0 + j] : 0.f;
        float xm = xv;
        if (drop.thr16) xm = (hv && ((mw[((int64_t)lay[j] * B + b) * W + (h >> 5)] >> (h & 31)) & 1u)) ? xd : 0.f;
        acc[j] = fmaf(g, xm, acc[j]);
        sb[j] += g;
      }
    }
#pragma unroll
    for (int j = 0; j < kBwdRows; ++j) {
      red[wave][j][lane] = acc[j];
      if (lane == 0) redb[wave][j] = sb[j];
    }
    __syncthreads();
    for (int j = wave; j < kBwdRows; j += 4) {
      const int r = r0 + j;
      if (r < R && hv) {
        const float v = (red[0][j][lane] + red[1][j][lane]) + (red[2][j][lane] + red[3][j][lane]);
        float* o = dWh + (int64_t)r * H + h;
        *o = accumulate ? *o + v : v;
      }
      if (r < R && lane == 0 && (bid % nh) == 0) {
        const float v = (redb[0][j] + redb[1][j]) + (redb[2][j] + redb[3][j]);
        dbh[r] = accumulate ? dbh[r] + v : v;
      }
    }
    return;
  }
  const int nD = ((B + kBwdRows - 1) / kBwdRows) * nh;
  if (bid < nW + nD) {
    // ---- dcls[b][h] = sum_r dz[b][r] * Wh[r][h] * dropmask_{layer(r)}(b, h) ----
    const int q = bid - nW, b0 = (q / nh) * kBwdRows, h = (q % nh) * 64 + lane;
    const bool hv = h < H;
    float acc[kBwdRows];
#pragma unroll
    for (int j = 0; j < kBwdRows; ++j) acc[j] = 0.f;
    for (int r = wave; r < R; r += 4) {
      const float w = hv ? Wh[(int64_t)r * H + h] : 0.f;
      const float wd = w * drop.scale;
      const int lay = lay_row[r];
#pragma unroll
      for (int j = 0; j < kBwdRows; ++j) {
        const int b = min(b0 + j, B - 1);
        const float g = dz[(int64_t)b * R + r];
        float wm = w;
        if (drop.thr16) wm = (hv && ((mw[((int64_t)lay * B + b) * W + (h >> 5)] >> (h & 31)) & 1u)) ? wd : 0.f;
        acc[j] = fmaf(g, wm, acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < kBwdRows; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    for (int j = wave; j < kBwdRows; j += 4) {
      const int b = b0 + j;
      if (b < B && hv) dcls[(int64_t)b * H + h] = (red[0][j][lane] + red[1][j][lane]) + (red[2][j][lane] + red[3][j][lane]);
    }
    return;
  }
  // ---- last block (launched only with loss_out): the four loss sums over the batch (fixed order) ----
  if (!loss_out) return;
  __shared__ float sm[16];
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) s += sample_loss[4 * b + k];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) loss_out[k] = s;
  }
}

// d(logits) from ARBITRARY upstream gradients of the three outputs (the autograd bridge: the reference's loop calls
// total_loss.backward() on whatever it built from top / bottoms / final, /root/reference/n_best_asr_bert.py:255-264):
//   final[b][bi] = top_t (single-bottom top) | top_t * s_j ;  s = softmax(head logits) ;  top = sigmoid(z_t)
// grid B, block 256: a wave per top label, lanes over the head columns (as heads_fwd_kernel).
__global__ __launch_bounds__(256) void heads_vjp_dz_kernel(const float* __restrict__ top, const float* __restrict__ bott,
                                                           const float* __restrict__ dtop, const float* __restrict__ dbott,
                                                           const float* __restrict__ dfin, const int32_t* __restrict__ bottom_off,
                                                           const int32_t* __restrict__ bottom_ids, const int32_t* __restrict__ head_row,
                                                           int n_top, int n_bottom, int R, float* __restrict__ dz) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const float* gf = dfin + (int64_t)b * n_bottom;
  for (int t = wave; t < n_top; t += nw) {
    const int o0 = bottom_off[t], nk = bottom_off[t + 1] - o0, hr = head_row[t];
    const float pt = top[(int64_t)b * n_top + t];
    float dpt = dtop[(int64_t)b * n_top + t];
    if (hr < 0) {
      dpt += gf[bottom_ids[o0]];
    } else {
      const float* sb = bott + (int64_t)b * (R - n_top) + (hr - n_top);
      const float* gb = dbott + (int64_t)b * (R - n_top) + (hr - n_top);
      float dot = 0.f, acc = 0.f;
      for (int j = lane; j < nk; j += 64) {
        const float s = sb[j], g = gf[bottom_ids[o0 + j]];
        const float ds = gb[j] + g * pt;
        dot += ds * s;
        acc += g * s;
      }
      dot = wave_sum(dot);
      dpt += wave_sum(acc);
      for (int j = lane; j < nk; j += 64) {
        const float s = sb[j];
        const float ds = gb[j] + gf[bottom_ids[o0 + j]] * pt;
        dz[(int64_t)b * R + hr + j] = s * (ds - dot);
      }
    }
    if (lane == 0) dz[(int64_t)b * R + t] = dpt * pt * (1.f - pt);
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
  const size_t smemF = ((size_t)H + 2 * (size_t)R) * sizeof(float) + (size_t)n_lay * ((H + 31) / 32) * sizeof(uint32_t) + (size_t)3 * (kFwdThreads / 64) * sizeof(float);
#define NB_HEADS_FWD(TT)                                                                                                              \
  heads_fwd_kernel<TT><<<B, kFwdThreads, smemF, st>>>((const TT*)hidden, cls_stride, Wh, bh, labels, ls->bottom_off, ls->bottom_ids,      \
                                                      ls->head_row, n_top, n_bottom, R, H, n_heads, n_lay, cls, top, bott, final_scores,  \
                                                      dz, sloss, mw, lay_row, d)
  if (dtype == NBEST_F32) NB_HEADS_FWD(float);
  else if (dtype == NBEST_BF16) NB_HEADS_FWD(bf16);
  else NB_CHECK(false, NBEST_ERR_DTYPE, "stc_heads: bad dtype %d", dtype);
#undef NB_HEADS_FWD
  NB_LAUNCH_CHECK();
  (void)logits;
  if (need_grad) {
    const int nh = (H + 63) / 64;
    const int nW = ((R + kBwdRows - 1) / kBwdRows) * nh, nD = ((B + kBwdRows - 1) / kBwdRows) * nh;
    heads_bwd_kernel<<<nW + nD + 1, 256, 0, st>>>(cls, dz, Wh, B, R, H, dWh, dbh, dcls, accumulate, mw, lay_row, d, sloss, loss_parts, nW);
  } else {
    loss_reduce_kernel<<<1, 256, 0, st>>>(sloss, B, loss_parts);
  }
  NB_LAUNCH_CHECK();
  return NBEST_OK;
}

// Backward of the heads for arbitrary upstream gradients (autograd bridge; hipabi.stc_heads_vjp): `ws` is the workspace a
// nbest_stc_heads call with the same (hidden, Wh, dropout seed) just used - it still holds the fp32 CLS rows, the layers' dropout bits
// and the row -> layer table.  dcls [B][H] is overwritten; dWh / dbh are overwritten unless accumulate.
extern "C" int nbest_stc_heads_vjp(const float* Wh, const nbest_label_space* ls, const float* top, const float* bott, const float* dtop,
                                   const float* dbott, const float* dfin, float* dcls, float* dWh, float* dbh, int B, int H, int accumulate,
                                   float drop_p, uint64_t seed, uint32_t drop_stream, void* ws, size_t ws_bytes, nbest_stream_t stream) {
  NB_CHECK(Wh && ls && top && bott && dtop && dbott && dfin && dcls && dWh && dbh && ws && B > 0 && H > 0, NBEST_ERR_ARG, "stc_heads_vjp: null pointer");
  const int R = ls->n_rows, n_top = ls->n_top, n_bottom = ls->n_bottom;
  NB_CHECK(ws_bytes >= nbest_heads_ws_bytes(B, R, H), NBEST_ERR_WORKSPACE, "stc_heads_vjp: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* cls = (float*)ws;
  float* dz = cls + (size_t)B * H + (size_t)B * R;
  float* sloss = dz + (size_t)B * R;
  uint32_t* mw = (uint32_t*)(sloss + (size_t)4 * B);
  int32_t* lay_row = (int32_t*)(mw + (size_t)(R + 1) * B * ((H + 31) / 32));
  const DropCfg d = make_drop(drop_p, seed, drop_stream);
  heads_vjp_dz_kernel<<<B, 256, 0, st>>>(top, bott, dtop, dbott, dfin, ls->bottom_off, ls->bottom_ids, ls->head_row, n_top, n_bottom, R, dz);
  NB_LAUNCH_CHECK();
  const int nh = (H + 63) / 64;
  const int nW = ((R + kBwdRows - 1) / kBwdRows) * nh, nD = ((B + kBwdRows - 1) / kBwdRows) * nh;
  heads_bwd_kernel<<<nW + nD, 256, 0, st>>>(cls, dz, Wh, B, R, H, dWh, dbh, dcls, accumulate, mw, lay_row, d, nullptr, nullptr, nW);
  NB_LAUNCH_CHECK();
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

namespace {
__global__ void stamp_kernel(int32_t* flag, int32_t value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" int nbest_stream_stamp(int32_t* flag, int32_t value, nbest_stream_t stream) {
  NB_CHECK(flag, NBEST_ERR_ARG, "stream_stamp: null pointer");
  stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(flag, value);
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
