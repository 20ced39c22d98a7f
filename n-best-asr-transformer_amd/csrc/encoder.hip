// The whole encoder stack behind one C-ABI call each way (forward / backward): every kernel of
// every layer is enqueued on the caller's stream from C++, so a training step costs a handful of
// Python->C transitions instead of ~400, and the sequence is hipGraph-capturable (no allocation, no
// synchronisation, no default-stream work inside).
//
// Activation stash `act` (written by forward, read by backward), T = dtype:
//   X[0..L]   [M][H] T      X[0] = embedding output, X[l+1] = output of layer l
//   emb_stats [M][2] f32
//   per layer: qkv [M][3H] T | ctx [M][H] T | lse [B*heads*S] f32 | r1 [M][H] T | st1 [M][2] f32 |
//              x1 [M][H] T | u = gelu'(pre-activation) [M][F] (fp32, or 8-bit fixed point in the bf16 path) | hact [M][F] T | r2 [M][H] T | st2 [M][2] f32
// Scratch `ws` (backward): dR | dRd | dB1 | dctx [M][H] T, dBig [M][F] T, dqkv [M][3H] T,
//              column-reduction partials, split-K slabs, embedding-backward buffer.
#include "common.h"

int nbest_internal_layernorm_fwd8(const void* x, const float* gamma, const float* beta, void* y, void* y8, float* stats,
                                  int64_t M, int H, float eps, int dtype, nbest_stream_t stream, const uint32_t* a_prev, uint32_t* a_new);
int nbest_internal_cast_bf16_to_fp8(const void* src, void* dst, int64_t n, const uint32_t* a_prev, uint32_t* a_new, hipStream_t st);
int nbest_internal_layernorm_bwd8(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                                  void* dx_drop, float* dgamma, float* dbeta, float* dbias, int64_t M, int H, int dtype,
                                  int accumulate, float drop_p, uint64_t seed, uint32_t drop_stream, void* ws,
                                  size_t ws_bytes, nbest_stream_t stream, Fp8Grad f8);
int nbest_internal_attention_bwd8(const void* qkv, const uint8_t* key_mask, const void* ctx, const void* dctx, const float* lse,
                                  void* dqkv, float* dbias, int accumulate, void* ws, size_t ws_bytes, int B, int S, int heads,
                                  int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, nbest_stream_t stream, Fp8Grad f8,
                                  const uint32_t* keep);
size_t nbest_internal_attention_keep_bytes(int B, int S, int heads);
int nbest_internal_amax_bf16(const void* x, int64_t n, uint32_t* out, hipStream_t st);
void nbest_internal_rowred_batch_begin();
void nbest_internal_rowred_batch_abort();
int nbest_internal_rowred_batch_flush(hipStream_t st);
int nbest_internal_attention_fwd8(const void* qkv, const uint8_t* key_mask, void* ctx, void* ctx8, float* lse, int B, int S, int heads,
                                  int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream, nbest_stream_t stream, uint32_t* keep,
                                  const uint32_t* a_prev, uint32_t* a_new);

namespace {

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct ActLayout {
  size_t esz, X, emb_stats, layer0, layer_stride;
  size_t o_qkv, o_ctx, o_lse, o_r1, o_st1, o_x1, o_u, o_hact, o_r2, o_st2;
  size_t o_x8, o_ctx8, o_x18, o_h8;   // fp8 forward ("fp8w"): e4m3 copies of the four GEMM inputs of the layer, kept for the fp8 weight gradients
  size_t o_keep, keep_bytes;          // bf16, S <= 256: attention-dropout keep words of the layer (forward -> backward)
  size_t total;
  int64_t M;
};

static ActLayout act_layout(const nbest_encoder_desc* d) {
  ActLayout a;
  a.esz = d->dtype == NBEST_BF16 ? 2 : 4;
  a.M = (int64_t)d->B * d->S;
  const size_t MH = al((size_t)a.M * d->H * a.esz), MF = al((size_t)a.M * d->F * a.esz), M3H = al((size_t)a.M * 3 * d->H * a.esz);
  const size_t st = al((size_t)a.M * 2 * sizeof(float)), lse = al((size_t)d->B * d->heads * d->S * sizeof(float));
  size_t o = 0;
  a.X = o; o += (size_t)(d->L + 1) * MH;
  a.emb_stats = o; o += st;
  a.layer0 = o;
  size_t p = 0;
  a.o_qkv = p; p += M3H;
  a.o_ctx = p; p += MH;
  a.o_lse = p; p += lse;
  a.o_r1 = p; p += MH;
  a.o_st1 = p; p += st;
  a.o_x1 = p; p += MH;
  a.o_u = p; p += (d->dtype == NBEST_BF16) ? al((size_t)a.M * d->F) : MF;   // GELU': 8 bits per element in the bf16 path
  a.o_hact = p; p += MF;
  a.o_r2 = p; p += MH;
  a.o_st2 = p; p += st;
  a.keep_bytes = (d->dtype == NBEST_BF16) ? nbest_internal_attention_keep_bytes(d->B, d->S, d->heads) : 0;
  a.o_keep = p; p += al(a.keep_bytes);
  a.o_x8 = a.o_ctx8 = a.o_x18 = a.o_h8 = 0;
  if (d->w8) {   // only the fp8 mode pays for them (+ (3 H + F) bytes per token and layer)
    const size_t MH8 = al((size_t)a.M * d->H), MF8 = al((size_t)a.M * d->F);
    a.o_x8 = p; p += MH8;
    a.o_ctx8 = p; p += MH8;
    a.o_x18 = p; p += MH8;
    a.o_h8 = p; p += MF8;
  }
  a.layer_stride = p;
  a.total = o + (size_t)d->L * p;
  return a;
}

struct WsLayout {
  size_t dR, dRd, dB1, dctx, dBig, dqkv, red, slab, slab_bytes, red_bytes, emb, emb_bytes, f8, f8_bytes, total;
};

static size_t max_splitk_bytes(const nbest_encoder_desc* d, int64_t M) {
  size_t mx = 0;
  const int64_t shapes[4][2] = {{3 * (int64_t)d->H, d->H}, {d->H, d->H}, {d->F, d->H}, {d->H, d->F}};
  for (int i = 0; i < 4; ++i) {
    nbest_gemm_args g = {};
    g.M = shapes[i][0]; g.N = shapes[i][1]; g.K = M; g.trans_a = g.trans_b = 1; g.epilogue = NBEST_EPI_F32_SPLITK;
    g.dtype = d->dtype;
    const size_t b = nbest_gemm_ws_bytes(&g);
    if (b > mx) mx = b;
    if (d->dtype == NBEST_BF16 && shapes[i][0] % 256 == 0 && shapes[i][1] % 256 == 0) {   // fp8 weight gradients: own split plan
      const size_t b8 = nbest_wgrad_fp8_ws_bytes(shapes[i][0], shapes[i][1], M);
      if (b8 > mx) mx = b8;
    }
  }
  if (d->dtype == NBEST_BF16) {   // the QKV + attention-output pair (nbest_wgrad_pair)
    nbest_gemm_args g1 = {}, g2 = {};
    g1.M = 3 * (int64_t)d->H; g2.M = d->H; g1.N = g2.N = d->H; g1.K = g2.K = M;
    g1.trans_a = g1.trans_b = g2.trans_a = g2.trans_b = 1; g1.epilogue = g2.epilogue = NBEST_EPI_F32_SPLITK; g1.dtype = g2.dtype = NBEST_BF16;
    const size_t bp = nbest_wgrad_pair_ws_bytes(&g1, &g2);
    if (bp > mx) mx = bp;
    const size_t bp8 = nbest_wgrad_fp8_pair_ws_bytes(3 * (int64_t)d->H, d->H, d->H, M);
    if (bp8 > mx) mx = bp8;
  }
  return mx;
}

// the attention-output weight gradient of a layer is issued together with the QKV gradient (nbest_wgrad_pair / nbest_wgrad_fp8_pair:
// 3 weight-gradient launches per layer instead of 4) when the pair fits one 256 x 256 split-K launch
static bool wgrad_paired(const nbest_encoder_desc* d, bool f8b) {
  if (d->dtype != NBEST_BF16) return false;
  if (f8b) return nbest_wgrad_fp8_pair_ws_bytes(3 * (int64_t)d->H, d->H, d->H, (int64_t)d->B * d->S) > 0;
  nbest_gemm_args g1 = {}, g2 = {};
  g1.M = 3 * (int64_t)d->H; g2.M = d->H; g1.N = g2.N = d->H; g1.K = g2.K = (int64_t)d->B * d->S;
  g1.trans_a = g1.trans_b = g2.trans_a = g2.trans_b = 1; g1.epilogue = g2.epilogue = NBEST_EPI_F32_SPLITK; g1.dtype = g2.dtype = NBEST_BF16;
  return nbest_wgrad_pair_ws_bytes(&g1, &g2) > 0;
}

static WsLayout ws_layout(const nbest_encoder_desc* d) {
  WsLayout w;
  const size_t esz = d->dtype == NBEST_BF16 ? 2 : 4;
  const int64_t M = (int64_t)d->B * d->S;
  const size_t MH = al((size_t)M * d->H * esz), MF = al((size_t)M * d->F * esz), M3H = al((size_t)M * 3 * d->H * esz);
  size_t o = 0;
  w.dR = o; o += MH;
  w.dRd = o; o += MH;
  w.dB1 = o; o += MH;
  w.dctx = o; o += MH;
  w.dBig = o; o += MF;
  w.dqkv = o; o += M3H;
  const int64_t maxN = d->F > 3 * d->H ? d->F : 3 * d->H;
  w.red_bytes = al(nbest_rowred_ws_bytes(M, maxN));
  {
    const size_t ab = al(nbest_attention_bwd_ws_bytes(d->B, d->S, d->heads));
    if (ab > w.red_bytes) w.red_bytes = ab;
    const size_t gb = al((size_t)((M + 127) / 128) * 2 * d->F * sizeof(float));   // fused column sums of the dU GEMM
    if (gb > w.red_bytes) w.red_bytes = gb;
  }
  w.red = o; o += 4 * w.red_bytes;      // four regions: the partial rows of a layer's four producers live until its one finalize
  w.slab_bytes = al(max_splitk_bytes(d, M));
  w.slab = o; o += w.slab_bytes;
  w.emb_bytes = al(nbest_embed_bwd_ws_bytes(M, d->H));
  w.emb = o; o += w.emb_bytes;
  // fp8 forward: e4m3 copies of the GEMM inputs of ONE layer (x | ctx | x1: [M][H] bytes each, gelu(u): [M][F] bytes)
  // (backward, fp8 dgrads: dQ|dK|dV copy over the first three blocks, FFN gradient copy over the fourth, a fifth [M][H] block)
  w.f8_bytes = (d->dtype == NBEST_BF16) ? 4 * al((size_t)M * d->H) + al((size_t)M * d->F) : 0;
  w.f8 = o; o += w.f8_bytes;
  w.total = o;
  return w;
}

// the backward of this pass runs its dgrad / wgrad GEMMs in fp8 (same answer in the forward, which then leaves out the
// bf16 tensors only a bf16 backward would read, and in the backward)
// the forward of this pass runs its GEMMs in fp8: needs the activation amax history (fp8_act; without one the pass is a calibration
// pass on the bf16 GEMMs that records it)
static bool fp8_forward_active(const nbest_encoder_desc* d) {
  return d->dtype == NBEST_BF16 && d->w8 && d->w8_inv_scale && d->fp8_act && d->aamax_prev;
}
static bool fp8_backward_active(const nbest_encoder_desc* d) {
  return fp8_forward_active(d) && d->fp8_bwd && d->w8t && d->gamax_prev && d->gamax_new;   // reads the forward's e4m3 activation copies
}

static int check_desc(const nbest_encoder_desc* d) {
  NB_CHECK(d && d->layers_host, NBEST_ERR_ARG, "encoder: null descriptor");
  NB_CHECK(d->dtype == NBEST_F32 || d->dtype == NBEST_BF16, NBEST_ERR_DTYPE, "encoder: bad dtype %d", d->dtype);
  NB_CHECK(d->B > 0 && d->S > 0 && d->L > 0 && d->heads > 0, NBEST_ERR_SHAPE, "encoder: bad shape");
  NB_CHECK(d->H == d->heads * 64, NBEST_ERR_SHAPE, "encoder: hidden %d != heads %d x 64", d->H, d->heads);
  // the reference never truncates (utils/bert_xlnet_inputs.py:87-94) and would index past the position table; fail up
  // front instead, before anything is enqueued (RoBERTa-family positions start at pad_id + 1)
  NB_CHECK(d->S + (d->pos_pad_id >= 0 ? (int)d->pos_pad_id + 1 : 0) <= d->max_pos, NBEST_ERR_SHAPE,
           "encoder: S=%d does not fit the position table (%d rows)", d->S, d->max_pos);
  NB_CHECK(d->S <= 512, NBEST_ERR_SHAPE, "encoder: S=%d > 512", d->S);
  if (d->dtype == NBEST_BF16)
    NB_CHECK(d->H % 128 == 0 && d->F % 128 == 0, NBEST_ERR_SHAPE, "encoder(bf16): H and F must be multiples of 128");
  if (d->w8) NB_CHECK(d->dtype == NBEST_BF16 && d->w8_inv_scale && d->H % 256 == 0 && d->H <= 1024 && d->F % 256 == 0, NBEST_ERR_SHAPE,
                      "encoder(fp8 forward): needs the bf16 path, inverse scales and H, F multiples of 256");
  return NBEST_OK;
}

struct Ptrs {
  const char* wts;   // matrices / tables, dtype
  const float* prm;  // fp32 master (biases, LayerNorm)
  size_t esz;
  const void* W(int64_t off) const { return wts + (size_t)off * esz; }
  const float* P(int64_t off) const { return prm + off; }
};

static int gemm(int dtype, const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                int64_t ldc, int ta, int tb, int epi, const float* bias, const void* R, int64_t ldr, void* U, int64_t ldu,
                void* ws, size_t ws_bytes, int accumulate, float drop_p, uint64_t seed, uint32_t stream_id, hipStream_t st,
                float* colsum_out = nullptr, const void* B_packed = nullptr) {
  nbest_gemm_args g = {};
  if (B_packed && !ta && !tb) { g.B_packed = B_packed; g.b_pack_bn = nbest_pack_bn(N); }
  g.colsum_out = colsum_out; g.colsum_accumulate = accumulate;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.R = R; g.U = U; g.ws = ws; g.ws_bytes = ws_bytes;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.ldu = ldu;
  g.trans_a = ta; g.trans_b = tb; g.epilogue = epi; g.dtype = dtype; g.accumulate = accumulate;
  g.drop_p = drop_p; g.drop_stream = stream_id; g.seed = seed;
  return nbest_gemm(&g, (nbest_stream_t)st);
}

#define RUN(x)          \
  do {                  \
    int rc__ = (x);     \
    if (rc__) return rc__; \
  } while (0)

}  // namespace

extern "C" size_t nbest_encoder_act_bytes(const nbest_encoder_desc* d) { return d ? act_layout(d).total : 0; }
extern "C" size_t nbest_encoder_ws_bytes(const nbest_encoder_desc* d) { return d ? ws_layout(d).total : 0; }
extern "C" int nbest_encoder_wgrad_launches_per_layer(const nbest_encoder_desc* d) {
  return d ? (wgrad_paired(d, fp8_backward_active(d)) ? 3 : 4) : 0;
}

extern "C" int nbest_encoder_forward(const nbest_encoder_desc* d, const void* wts, const float* prm, const int64_t* ids,
                                     const int64_t* seg, const int64_t* pos, const uint8_t* key_mask, void* act, size_t act_bytes,
                                     void* ws, size_t ws_bytes, void** hidden_out, nbest_stream_t stream) {
  RUN(check_desc(d));
  NB_CHECK(wts && prm && ids && pos && key_mask && act, NBEST_ERR_ARG, "encoder_forward: null pointer");
  const ActLayout a = act_layout(d);
  NB_CHECK(act_bytes >= a.total, NBEST_ERR_WORKSPACE, "encoder_forward: activation stash too small (%zu < %zu)", act_bytes, a.total);
  hipStream_t st = (hipStream_t)stream;
  const bool f8 = fp8_forward_active(d);
  const bool arec = d->w8 && d->aamax_new && d->dtype == NBEST_BF16;     // record the activation amax of this pass (fp8 or calibration)
  auto AP = [&](int idx) -> const uint32_t* { return f8 ? d->aamax_prev + idx : nullptr; };
  auto AN = [&](int idx) -> uint32_t* { return arec ? d->aamax_new + (int64_t)idx * NBEST_AMAX_TENSOR_WORDS : nullptr; };   // slot block of tensor idx
  const WsLayout wl = ws_layout(d);
  if (f8) NB_CHECK(ws && ws_bytes >= wl.total, NBEST_ERR_WORKSPACE, "encoder_forward(fp8): workspace too small (%zu < %zu)", ws_bytes, wl.total);
  const Ptrs P{(const char*)wts, prm, a.esz};
  char* A = (char*)act;
  const int64_t M = a.M;
  const int H = d->H, F = d->F, dt = d->dtype;
  const size_t MH = al((size_t)M * H * a.esz);
  auto PK = [&](int64_t off) -> const void* { return (d->wpk && dt == NBEST_BF16) ? (const void*)((const char*)d->wpk + off * 2) : nullptr; };
  auto X = [&](int l) { return (void*)(A + a.X + (size_t)l * MH); };
  const uint32_t sb = d->drop_stream_base;

  RUN(nbest_embed_ln_fwd(ids, seg, pos, P.W(d->off_word), P.W(d->off_type), P.W(d->off_pos), P.P(d->off_emb_ln_g),
                         P.P(d->off_emb_ln_b), X(0), (float*)(A + a.emb_stats), M, H, d->ln_eps, dt, d->hidden_drop, d->seed, sb, st));
  // fp8 forward: the four GEMMs of a layer on the block-scaled fp8 MFMA; their A operands are e4m3 copies in `ws`
  // (kept per layer in the activation stash: the fp8 weight gradients of the backward read them again)
  auto L8 = [&](int l, size_t off) -> uint8_t* { return f8 ? (uint8_t*)(A + a.layer0 + (size_t)l * a.layer_stride + off) : nullptr; };
  auto gemm8 = [&](const uint8_t* A8, int64_t w_off, int mat, void* Cout, int64_t N, int64_t K, int epi, const float* bias, const void* R,
                   void* U, uint8_t* C8, float drop_p, uint32_t stream_id) -> int {
    nbest_gemm_fp8_args g = {};
    g.A = A8; g.B = (const uint8_t*)d->w8 + w_off; g.C = Cout; g.bias = bias; g.R = R; g.U = U; g.C8 = C8;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N; g.ldu = N; g.ldc8 = N;
    if (d->w8p) { g.B_packed = (const uint8_t*)d->w8p + w_off; g.b_pack_bn = nbest_pack_bn_fp8(N, K); }
    g.epilogue = epi; g.out_scale = 1.f; g.out_scale_dev = d->w8_inv_scale + mat; g.drop_p = drop_p; g.drop_stream = stream_id; g.seed = d->seed;
    g.a_amax = d->aamax_prev + mat;         // the A operand's delayed scale (activation index = matrix index: 4 l + {x, ctx, x1, gelu})
    if (epi == NBEST_EPI_BIAS_GELU) { g.c8_amax_prev = d->aamax_prev + mat + 1; g.c8_amax_new = d->aamax_new ? d->aamax_new + (int64_t)(mat + 1) * NBEST_AMAX_TENSOR_WORDS : nullptr; }
    return nbest_gemm_fp8(&g, stream);
  };
  // calibration pass (fp8 mode without an activation history): bf16 GEMMs, the amax of the four GEMM inputs of every layer recorded
  auto calib = [&](const void* t, int64_t n, int idx) -> int { return (arec && !f8) ? nbest_internal_amax_bf16(t, n, d->aamax_new + (int64_t)idx * NBEST_AMAX_TENSOR_WORDS, st) : NBEST_OK; };
  for (int l = 0; l < d->L; ++l) {
    const nbest_layer_offsets& o = d->layers_host[l];
    char* Lb = A + a.layer0 + (size_t)l * a.layer_stride;
    void* qkv = Lb + a.o_qkv; void* ctx = Lb + a.o_ctx; float* lse = (float*)(Lb + a.o_lse);
    void* r1 = Lb + a.o_r1; float* st1 = (float*)(Lb + a.o_st1); void* x1 = Lb + a.o_x1;
    void* u = Lb + a.o_u; void* hact = Lb + a.o_hact; void* r2 = Lb + a.o_r2; float* st2 = (float*)(Lb + a.o_st2);
    const uint32_t s0 = sb + 1 + 4 * l;
    uint8_t* x8 = L8(l, a.o_x8); uint8_t* ctx8 = L8(l, a.o_ctx8); uint8_t* x18 = L8(l, a.o_x18); uint8_t* h8 = L8(l, a.o_h8);
    uint8_t* x8_next = (l + 1 < d->L) ? L8(l + 1, a.o_x8) : nullptr;   // the last LayerNorm's copy has no reader
    // QKV projection: [M,H] x [3H,H]^T + b
    if (f8) {
      if (l == 0) RUN(nbest_internal_cast_bf16_to_fp8(X(0), x8, M * H, AP(0), AN(0), st));   // later layers: written by the previous layer's LayerNorm
      RUN(gemm8(x8, o.wqkv, 4 * l + 0, qkv, 3 * H, H, NBEST_EPI_BIAS, P.P(o.bqkv), nullptr, nullptr, nullptr, 0.f, 0));
    } else
    RUN(gemm(dt, X(l), P.W(o.wqkv), qkv, M, 3 * H, H, H, H, 3 * H, 0, 0, NBEST_EPI_BIAS, P.P(o.bqkv), nullptr, 0, nullptr, 0,
             nullptr, 0, 0, 0.f, 0, 0, st, nullptr, PK(o.wqkv)));
    uint32_t* keepw = (a.keep_bytes && d->attn_drop > 0.f) ? (uint32_t*)(Lb + a.o_keep) : nullptr;
    RUN(calib(X(l), M * H, 4 * l + 0));
    RUN(nbest_internal_attention_fwd8(qkv, key_mask, ctx, ctx8, lse, d->B, d->S, d->heads, 64, dt, d->attn_drop, d->seed, s0 + 0, stream, keepw,
                                      AP(4 * l + 1), f8 ? AN(4 * l + 1) : nullptr));
    RUN(calib(ctx, M * H, 4 * l + 1));
    // attention output projection + dropout + residual, then LayerNorm
    if (f8) {
      RUN(gemm8(ctx8, o.wo, 4 * l + 1, r1, H, H, NBEST_EPI_BIAS_DROP_RES, P.P(o.bo), X(l), nullptr, nullptr, d->hidden_drop, s0 + 1));
    } else
    RUN(gemm(dt, ctx, P.W(o.wo), r1, M, H, H, H, H, H, 0, 0, NBEST_EPI_BIAS_DROP_RES, P.P(o.bo), X(l), H, nullptr, 0, nullptr, 0, 0,
             d->hidden_drop, d->seed, s0 + 1, st, nullptr, PK(o.wo)));
    RUN(nbest_internal_layernorm_fwd8(r1, P.P(o.ln1_g), P.P(o.ln1_b), x1, x18, st1, M, H, d->ln_eps, dt, stream, AP(4 * l + 2), f8 ? AN(4 * l + 2) : nullptr));
    RUN(calib(x1, M * H, 4 * l + 2));
    // FFN up + bias + GELU (GELU' of the pre-activation kept for the backward)
    if (f8) {
      // (bf16 gelu(u) has one reader, the bf16 FFN-down weight gradient: not written when the backward runs in fp8)
      RUN(gemm8(x18, o.w1, 4 * l + 2, fp8_backward_active(d) ? nullptr : hact, F, H, NBEST_EPI_BIAS_GELU, P.P(o.b1), nullptr, u, h8, 0.f, 0));
    } else
    RUN(gemm(dt, x1, P.W(o.w1), hact, M, F, H, H, H, F, 0, 0, NBEST_EPI_BIAS_GELU, P.P(o.b1), nullptr, 0, u, F, nullptr, 0, 0, 0.f,
             0, 0, st, nullptr, PK(o.w1)));
    // FFN down + dropout + residual, then LayerNorm
    if (f8) {
      RUN(gemm8(h8, o.w2, 4 * l + 3, r2, H, F, NBEST_EPI_BIAS_DROP_RES, P.P(o.b2), x1, nullptr, nullptr, d->hidden_drop, s0 + 2));
    } else
    RUN(gemm(dt, hact, P.W(o.w2), r2, M, H, F, F, F, H, 0, 0, NBEST_EPI_BIAS_DROP_RES, P.P(o.b2), x1, H, nullptr, 0, nullptr, 0, 0,
             d->hidden_drop, d->seed, s0 + 2, st, nullptr, PK(o.w2)));
    RUN(calib(hact, M * F, 4 * l + 3));
    RUN(nbest_internal_layernorm_fwd8(r2, P.P(o.ln2_g), P.P(o.ln2_b), X(l + 1), x8_next, st2, M, H, d->ln_eps, dt, stream,
                                      x8_next ? AP(4 * l + 4) : nullptr, (f8 && x8_next) ? AN(4 * l + 4) : nullptr));
  }
  if (hidden_out) *hidden_out = X(d->L);
  return NBEST_OK;
}

extern "C" int nbest_encoder_backward(const nbest_encoder_desc* d, const void* wts, const void* wts_t, const float* prm, float* grad,
                                      const int64_t* ids, const int64_t* seg, const int64_t* pos, const uint8_t* key_mask,
                                      void* act, size_t act_bytes, void* dhidden, void* ws, size_t ws_bytes, int accumulate,
                                      int layer_begin, int layer_end, int with_embeddings, nbest_stream_t stream) {
  RUN(check_desc(d));
  NB_CHECK(0 <= layer_begin && layer_begin <= layer_end && layer_end <= d->L, NBEST_ERR_ARG, "encoder_backward: bad layer range");
  NB_CHECK(wts && prm && grad && ids && pos && key_mask && act && dhidden && ws, NBEST_ERR_ARG, "encoder_backward: null pointer");
  NB_CHECK(!with_embeddings || d->word_perm, NBEST_ERR_ARG, "encoder_backward: desc.word_perm (stable argsort of this pass's ids) is required");
  const ActLayout a = act_layout(d);
  const WsLayout w = ws_layout(d);
  NB_CHECK(act_bytes >= a.total, NBEST_ERR_WORKSPACE, "encoder_backward: activation stash too small");
  NB_CHECK(ws_bytes >= w.total, NBEST_ERR_WORKSPACE, "encoder_backward: workspace too small (%zu < %zu)", ws_bytes, w.total);
  hipStream_t st = (hipStream_t)stream;
  const Ptrs P{(const char*)wts, prm, a.esz};
  // dgrad operand: with a transposed weight arena both GEMM operands are k-contiguous (B given as [N][K])
  const bool wt = (wts_t != nullptr);
  const Ptrs PT{(const char*)(wt ? wts_t : wts), prm, a.esz};
  const int tbd = wt ? 0 : 1;
  auto PKT = [&](int64_t off) -> const void* { return (wt && d->wpkt && d->dtype == NBEST_BF16) ? (const void*)((const char*)d->wpkt + off * 2) : nullptr; };
  char* A = (char*)act;
  char* W = (char*)ws;
  const int64_t M = a.M;
  const int H = d->H, F = d->F, dt = d->dtype;
  const size_t MH = al((size_t)M * H * a.esz);
  auto X = [&](int l) { return (void*)(A + a.X + (size_t)l * MH); };
  auto G = [&](int64_t off) { return grad + off; };
  void* dA = dhidden;
  void* dR = W + w.dR;
  const bool hdrop = d->hidden_drop > 0.f;
  void* dRd = hdrop ? (void*)(W + w.dRd) : dR;
  void* dB1 = W + w.dB1; void* dctx = W + w.dctx; void* dBig = W + w.dBig; void* dqkv = W + w.dqkv;
  void* red = W + w.red; void* slab = W + w.slab;
  void* red1 = W + w.red + w.red_bytes; void* red2 = W + w.red + 2 * w.red_bytes; void* red3 = W + w.red + 3 * w.red_bytes;
  const uint32_t sb = d->drop_stream_base;
  // optional in-step timing of the weight-gradient GEMMs (see nbest_encoder_desc::wgrad_events)
  const bool f8b = fp8_backward_active(d);
  const bool paired = wgrad_paired(d, f8b);
  int ev_i = (paired ? 3 : 4) * (d->L - layer_end);
  auto stamp = [&](int which) {
    if (d->wgrad_events && 2 * ev_i + which < d->wgrad_events_n) (void)hipEventRecord((hipEvent_t)d->wgrad_events[2 * ev_i + which], st);
    ev_i += which;
  };

  // fp8 dgrads (descriptor: w8t, gamax_prev / gamax_new, fp8_bwd): the gradient amax of every dgrad operand is recorded in
  // every pass; with a history (fp8_bwd) the producers also write e4m3 copies and the four dgrad GEMMs of a layer run in fp8
  const bool rec = d->gamax_new && dt == NBEST_BF16;
  uint8_t* dqkv8 = f8b ? (uint8_t*)W + w.f8 : nullptr;                       // [M][3H]
  uint8_t* dBig8 = f8b ? dqkv8 + 3 * al((size_t)M * H) : nullptr;            // [M][F]
  uint8_t* dRd8 = f8b ? dBig8 + al((size_t)M * F) : nullptr;                 // [M][H]
  auto fg = [&](uint8_t* out8, int idx) -> Fp8Grad {
    if (!rec) return Fp8Grad{nullptr, nullptr, nullptr};
    return Fp8Grad{f8b ? out8 : nullptr, f8b ? d->gamax_prev + idx : nullptr, d->gamax_new + (int64_t)idx * NBEST_AMAX_TENSOR_WORDS};
  };
  auto dgrad8 = [&](const uint8_t* A8, int a_idx, int64_t w_off, int mat, void* Cout, int64_t N, int64_t K, int epi, const void* R,
                    void* U, uint8_t* C8, int c_idx, float* colsum) -> int {
    nbest_gemm_fp8_args g = {};
    g.A = A8; g.B = (const uint8_t*)d->w8t + w_off; g.C = Cout; g.R = R; g.U = U; g.C8 = C8;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N; g.ldu = N; g.ldc8 = N;
    if (d->w8tp) { g.B_packed = (const uint8_t*)d->w8tp + w_off; g.b_pack_bn = nbest_pack_bn_fp8(N, K); }
    g.epilogue = epi; g.out_scale = 1.f; g.out_scale_dev = d->w8_inv_scale + mat; g.a_amax = d->gamax_prev + a_idx;
    if (c_idx >= 0) { g.c8_amax_prev = d->gamax_prev + c_idx; g.c8_amax_new = d->gamax_new + (int64_t)c_idx * NBEST_AMAX_TENSOR_WORDS; }
    g.colsum_out = colsum; g.colsum_accumulate = accumulate; g.ws = red1; g.ws_bytes = w.red_bytes;
    return nbest_gemm_fp8(&g, stream);
  };

  struct BatchGuard { ~BatchGuard() { nbest_internal_rowred_batch_abort(); } } batch_guard;   // an error return mid-layer must not leave it open
  for (int l = layer_end - 1; l >= layer_begin; --l) {
    const nbest_layer_offsets& o = d->layers_host[l];
    char* Lb = A + a.layer0 + (size_t)l * a.layer_stride;
    void* qkv = Lb + a.o_qkv; void* ctx = Lb + a.o_ctx; float* lse = (float*)(Lb + a.o_lse);
    void* r1 = Lb + a.o_r1; float* st1 = (float*)(Lb + a.o_st1); void* x1 = Lb + a.o_x1;
    void* u = Lb + a.o_u; void* hact = Lb + a.o_hact; void* r2 = Lb + a.o_r2; float* st2 = (float*)(Lb + a.o_st2);
    const uint32_t s0 = sb + 1 + 4 * l;
    nbest_internal_rowred_batch_begin();    // the layer's four bias / LayerNorm-parameter reductions: one finalize launch at its end
    const uint8_t* x8 = (const uint8_t*)(Lb + a.o_x8); const uint8_t* ctx8 = (const uint8_t*)(Lb + a.o_ctx8);
    const uint8_t* x18 = (const uint8_t*)(Lb + a.o_x18); const uint8_t* h8 = (const uint8_t*)(Lb + a.o_h8);
    // LN2 backward: dR (residual branch), dRd (dense branch, under the dropout mask), db2
    // (with fp8 dgrads / wgrads the bf16 forms of dRd, dBig and dqkv have no reader: only their e4m3 copies are written)
    RUN(nbest_internal_layernorm_bwd8(dA, r2, st2, P.P(o.ln2_g), dR, (hdrop && !f8b) ? dRd : nullptr, G(o.ln2_g), G(o.ln2_b), G(o.b2), M, H, dt,
                                      accumulate, d->hidden_drop, d->seed, s0 + 2, red, w.red_bytes, stream, fg(dRd8, 4 * l + 0)));   // partial rows: region 0
    // FFN-down: dgrad fused with GELU' -> dU ; wgrad
    // (the FFN-up bias gradient = column sums of dU is fused into this epilogue)
    if (f8b) {
      RUN(dgrad8(dRd8, 4 * l + 0, o.w2, 4 * l + 3, nullptr, F, H, NBEST_EPI_DGELU, nullptr, u, dBig8, 4 * l + 1, G(o.b1)));
    } else {
      RUN(gemm(dt, dRd, PT.W(o.w2), dBig, M, F, H, H, wt ? H : F, F, 0, tbd, NBEST_EPI_DGELU, nullptr, nullptr, 0, u, F, red1, w.red_bytes, accumulate,
               0.f, 0, 0, st, G(o.b1), PKT(o.w2)));
      if (rec) RUN(nbest_internal_amax_bf16(dBig, M * F, d->gamax_new + (int64_t)(4 * l + 1) * NBEST_AMAX_TENSOR_WORDS, st));   // calibration pass: this producer is a bf16 kernel
    }
    stamp(0);
    if (f8b) RUN(nbest_wgrad_fp8(dRd8, h8, G(o.w2), H, F, M, H, F, F, d->gamax_prev + 4 * l + 0, d->aamax_prev + 4 * l + 3, accumulate, slab, w.slab_bytes, stream));
    else RUN(gemm(dt, dRd, hact, G(o.w2), H, F, M, H, F, F, 1, 1, NBEST_EPI_F32_SPLITK, nullptr, nullptr, 0, nullptr, 0, slab, w.slab_bytes,
                  accumulate, 0.f, 0, 0, st));
    stamp(1);
    // FFN-up: dgrad + residual gradient ; wgrad
    if (f8b) RUN(dgrad8(dBig8, 4 * l + 1, o.w1, 4 * l + 2, dB1, H, F, NBEST_EPI_RES, dR, nullptr, nullptr, -1, nullptr));
    else RUN(gemm(dt, dBig, PT.W(o.w1), dB1, M, H, F, F, wt ? F : H, H, 0, tbd, NBEST_EPI_RES, nullptr, dR, H, nullptr, 0, nullptr, 0, 0, 0.f, 0, 0, st, nullptr, PKT(o.w1)));
    stamp(0);
    if (f8b) RUN(nbest_wgrad_fp8(dBig8, x18, G(o.w1), F, H, M, F, H, H, d->gamax_prev + 4 * l + 1, d->aamax_prev + 4 * l + 2, accumulate, slab, w.slab_bytes, stream));
    else RUN(gemm(dt, dBig, x1, G(o.w1), F, H, M, F, H, H, 1, 1, NBEST_EPI_F32_SPLITK, nullptr, nullptr, 0, nullptr, 0, slab, w.slab_bytes,
                  accumulate, 0.f, 0, 0, st));
    stamp(1);
    // LN1 backward
    RUN(nbest_internal_layernorm_bwd8(dB1, r1, st1, P.P(o.ln1_g), dR, (hdrop && !f8b) ? dRd : nullptr, G(o.ln1_g), G(o.ln1_b), G(o.bo), M, H, dt,
                                      accumulate, d->hidden_drop, d->seed, s0 + 1, red2, w.red_bytes, stream, fg(dRd8, 4 * l + 2)));
    // attention output projection: dgrad ; wgrad
    if (f8b) RUN(dgrad8(dRd8, 4 * l + 2, o.wo, 4 * l + 1, dctx, H, H, NBEST_EPI_NONE, nullptr, nullptr, nullptr, -1, nullptr));
    else RUN(gemm(dt, dRd, PT.W(o.wo), dctx, M, H, H, H, H, H, 0, tbd, NBEST_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0.f, 0, 0, st, nullptr, PKT(o.wo)));
    // (bf16: this layer's dRd and ctx stay untouched until the next layer's LayerNorm backward - the gradient is issued below, with QKV's)
    if (!paired) {
      stamp(0);
      if (f8b) RUN(nbest_wgrad_fp8(dRd8, ctx8, G(o.wo), H, H, M, H, H, H, d->gamax_prev + 4 * l + 2, d->aamax_prev + 4 * l + 1, accumulate, slab, w.slab_bytes, stream));
      else RUN(gemm(dt, dRd, ctx, G(o.wo), H, H, M, H, H, H, 1, 1, NBEST_EPI_F32_SPLITK, nullptr, nullptr, 0, nullptr, 0, slab, w.slab_bytes,
                    accumulate, 0.f, 0, 0, st));
      stamp(1);
    }
    // attention backward -> dqkv ; QKV bias gradient
    RUN(nbest_internal_attention_bwd8(qkv, key_mask, ctx, dctx, lse, f8b ? nullptr : dqkv, G(o.bqkv), accumulate, red3, w.red_bytes, d->B, d->S, d->heads, 64,
                                      dt, d->attn_drop, d->seed, s0 + 0, stream, fg(dqkv8, 4 * l + 3),
                                      (a.keep_bytes && d->attn_drop > 0.f) ? (const uint32_t*)(Lb + a.o_keep) : nullptr));
    // QKV projection: dgrad + residual gradient -> gradient wrt the layer input ; wgrad
    if (f8b) RUN(dgrad8(dqkv8, 4 * l + 3, o.wqkv, 4 * l + 0, dA, H, 3 * H, NBEST_EPI_RES, dR, nullptr, nullptr, -1, nullptr));
    else RUN(gemm(dt, dqkv, PT.W(o.wqkv), dA, M, H, 3 * H, 3 * H, wt ? 3 * H : H, H, 0, tbd, NBEST_EPI_RES, nullptr, dR, H, nullptr, 0, nullptr, 0, 0, 0.f, 0, 0, st, nullptr, PKT(o.wqkv)));
    stamp(0);
    if (f8b && paired)
      RUN(nbest_wgrad_fp8_pair(dqkv8, x8, G(o.wqkv), 3 * H, 3 * H, H, H, d->gamax_prev + 4 * l + 3, d->aamax_prev + 4 * l + 0, dRd8, ctx8, G(o.wo), H, H,
                               H, H, d->gamax_prev + 4 * l + 2, d->aamax_prev + 4 * l + 1, H, M, accumulate, slab, w.slab_bytes, stream));
    else if (f8b) RUN(nbest_wgrad_fp8(dqkv8, x8, G(o.wqkv), 3 * H, H, M, 3 * H, H, H, d->gamax_prev + 4 * l + 3, d->aamax_prev + 4 * l + 0, accumulate, slab,
                                      w.slab_bytes, stream));
    else if (paired) {
      nbest_gemm_args g1 = {}, g2 = {};
      g1.A = dqkv; g1.B = X(l); g1.C = G(o.wqkv); g1.M = 3 * H; g1.lda = 3 * H;
      g2.A = dRd; g2.B = ctx; g2.C = G(o.wo); g2.M = H; g2.lda = H;
      g1.N = g2.N = H; g1.K = g2.K = M; g1.ldb = g2.ldb = g1.ldc = g2.ldc = H;
      g1.trans_a = g1.trans_b = g2.trans_a = g2.trans_b = 1; g1.epilogue = g2.epilogue = NBEST_EPI_F32_SPLITK;
      g1.dtype = g2.dtype = dt; g1.accumulate = g2.accumulate = accumulate; g1.ws = slab; g1.ws_bytes = w.slab_bytes;
      RUN(nbest_wgrad_pair(&g1, &g2, stream));
    } else RUN(gemm(dt, dqkv, X(l), G(o.wqkv), 3 * H, H, M, 3 * H, H, H, 1, 1, NBEST_EPI_F32_SPLITK, nullptr, nullptr, 0, nullptr, 0, slab,
                  w.slab_bytes, accumulate, 0.f, 0, 0, st));
    stamp(1);
    RUN(nbest_internal_rowred_batch_flush(st));
  }
  if (!with_embeddings) return NBEST_OK;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(G(d->off_word), 0, (size_t)d->vocab * H * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(G(d->off_pos), 0, (size_t)d->max_pos * H * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(G(d->off_type), 0, (size_t)d->n_types * H * sizeof(float), st);
    NB_CHECK(e == hipSuccess, NBEST_ERR_LAUNCH, "encoder_backward: memset failed: %s", hipGetErrorString(e));
  }
  NB_CHECK(d->word_perm, NBEST_ERR_ARG, "encoder_backward: desc.word_perm (stable argsort of this pass's ids) is required");
  RUN(nbest_embed_ln_bwd(ids, seg, pos, d->word_perm, P.W(d->off_word), P.W(d->off_type), P.W(d->off_pos), P.P(d->off_emb_ln_g),
                         (const float*)(A + a.emb_stats), dA, G(d->off_word), G(d->off_type), G(d->off_pos), G(d->off_emb_ln_g),
                         G(d->off_emb_ln_b), d->B, d->S, H, d->n_types, dt, d->word_pad_id, d->pos_pad_id, accumulate, accumulate,
                         d->hidden_drop, d->seed, sb, W + w.emb, w.emb_bytes, stream));
  return NBEST_OK;
}
