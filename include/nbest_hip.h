/* nbest_hip.h - C ABI of libnbest_hip.so: the MI355X (gfx950) fine-tuning hot path of
 * N-Best-ASR-Transformer.
 *
 * The reference has NO native / FFI / plugin interface: its only seam is the Python
 * object injected at /root/reference/models/model.py:19 and called at :43-45,54-56.  This header is
 * therefore the boundary a maintainer binds from Python with ctypes (INTEGRATION.md shows the stub);
 * each entry point cites the reference (or third-party) code whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller unless the
 *     name ends in _host; the library never allocates, never synchronises, never touches the default
 *     stream: work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - returns NBEST_OK (0) or a negative NBEST_ERR_*; nbest_last_error() gives the message
 *     (thread-local).  No global mutable state and no environment variables: entry points are re-entrant and the
 *     shipped library has one code path per shape (experiment switches exist only in `make diag` builds).
 *   - dtype: NBEST_F32 or NBEST_BF16 = type of activations, weight matrices and embedding tables.
 *     Biases, LayerNorm parameters, statistics, losses, gradients of parameters and optimizer state
 *     are always fp32.
 *   - matrices are row-major; "M" is the number of token rows B*S.
 *   - dropout: (p, seed, stream_id) select a counter-based mask that forward and backward
 *     regenerate identically; p = 0 disables it (parity runs).
 */
#ifndef NBEST_HIP_H
#define NBEST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBEST_ABI_VERSION 1

enum { NBEST_OK = 0, NBEST_ERR_ARG = -1, NBEST_ERR_SHAPE = -2, NBEST_ERR_DTYPE = -3, NBEST_ERR_ALIGN = -4,
       NBEST_ERR_LAUNCH = -5, NBEST_ERR_WORKSPACE = -6 };
enum { NBEST_F32 = 0, NBEST_BF16 = 1 };

typedef void* nbest_stream_t; /* hipStream_t */

int nbest_version(void);
/* copies the calling thread's last error message into buf (NUL-terminated); returns its length */
int nbest_last_error(char* buf, size_t n);

/* ---------------------------------------------------------------------------------------------
 * K1  embedding gather + LayerNorm (+dropout)
 * replaces transformers BertEmbeddings / XLMRobertaEmbeddings (installed modeling_bert.py:53-108),
 * reached from /root/reference/models/model.py:43-45.
 *   out[m,:] = drop(LN(word[ids[m]] + type[seg[m]] + ptab[pos[m]]))     stats[m] = {mean, rstd}
 * seg may be NULL (all zeros).  pos is explicit (BERT: arange; XLM-R: pad-offset cumsum).          */
int nbest_embed_ln_fwd(const int64_t* ids, const int64_t* seg, const int64_t* pos, const void* word,
                       const void* type, const void* ptab, const float* gamma, const float* beta, void* out,
                       float* stats, int64_t M, int H, float eps, int dtype, float drop_p, uint64_t seed,
                       uint32_t drop_stream, nbest_stream_t stream);
/* backward: LN backward on dout, then the sums into the fp32 table gradients - the index_add of the installed BertEmbeddings
 * backward behind /root/reference/models/model.py:43-45 - WITHOUT float atomics: a segmented reduce over the tokens sorted by
 * word id, every addition in a fixed order (bit-reproducible run to run).
 *   perm [B*S] int32 (REQUIRED): token indices sorted by ids, ties in ascending token index (any STABLE argsort of ids; the
 *        data loader builds it next to ids - nbest_amd/trainer.py EncodedSplit.host_batch, bench.py).
 * Rows word_pad_id of dword and pos_pad_id of dptab receive nothing (nn.Embedding padding_idx; pass -1 for "no padding row").
 * tables_accumulate == 0: the table rows the batch touches are OVERWRITTEN (the caller has zeroed the rest of dword / dptab /
 * dtype_tab); != 0: they are added to.  dgamma / dbeta are overwritten unless accumulate.  Deterministic for inputs whose
 * position ids are the same in every sequence up to padding rows and whose token types are 0 / 1 (BERT, the RoBERTa family);
 * other rows fall back to float atomics on dptab / dtype_tab.  M = B*S < 2^31.  ws: >= nbest_embed_bwd_ws_bytes(M, H) bytes. */
int nbest_embed_ln_bwd(const int64_t* ids, const int64_t* seg, const int64_t* pos, const int32_t* perm, const void* word,
                       const void* type, const void* ptab, const float* gamma, const float* stats,
                       const void* dout, float* dword, float* dtype_tab, float* dptab, float* dgamma,
                       float* dbeta, int B, int S, int H, int n_types, int dtype, int64_t word_pad_id,
                       int64_t pos_pad_id, int accumulate, int tables_accumulate, float drop_p, uint64_t seed,
                       uint32_t drop_stream, void* ws, size_t ws_bytes, nbest_stream_t stream);
size_t nbest_embed_bwd_ws_bytes(int64_t M, int64_t H);

/* Sparse exchange of word-embedding gradient rows between data-parallel ranks (new functionality: the reference is single-process,
 * /root/reference/n_best_asr_bert.py:232-294; the table is the nn.Embedding of the installed modeling_bert.py:154-177).  With a
 * 250 002-row vocabulary only the rows of the tokens in a rank's shard are non-zero, so ranks exchange (row ids, row values):
 *   nbest_rows_gather: slot i < n_rows of ids_out / vals_out takes table row rows[i] ([.][H] fp32, H % 4 == 0); slots n_rows .. cap-1
 *                      are padding (id -1, zeros), so every rank sends the same `cap` slots;
 *   nbest_rows_zero:   table[ids[i]][:] = 0;
 *   nbest_rows_add:    table[ids[i]][:] += vals[i][:], ids unique within one call, negative ids skipped; no atomics - called once per
 *                      rank's block in rank order, every replica performs the same additions in the same order.                    */
int nbest_rows_gather(const float* table, const int64_t* rows, int64_t n_rows, int64_t cap, int64_t* ids_out, float* vals_out, int H,
                      nbest_stream_t stream);
int nbest_rows_zero(float* table, const int64_t* ids, int64_t n, int H, nbest_stream_t stream);
int nbest_rows_add(float* table, const int64_t* ids, const float* vals, int64_t n, int H, nbest_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2/K4/K6  dense GEMM on MFMA with fused epilogues
 * replaces nn.Linear inside BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput
 * (installed modeling_bert.py:154-177, 282-293, 325-351) and their autograd backward.
 *   C[M,N] = epi( op(A)[M,K] . op(B)[K,N] )
 *   trans_a = 0: A stored [M][K] (lda)          trans_a = 1: A stored [K][M]
 *   trans_b = 0: B stored [N][K] (torch Linear)  trans_b = 1: B stored [K][N]
 * N must be a multiple of 128, K of 64 unless trans_a (then K is free: token dimension); lda/ldb/ldc
 * multiples of 8 elements; pointers 16-byte aligned.                                                */
enum {
  NBEST_EPI_NONE = 0,          /* C = acc                                                    */
  NBEST_EPI_BIAS = 1,          /* C = acc + bias[n]                                          */
  NBEST_EPI_BIAS_GELU = 2,     /* u = acc + bias[n]; C = gelu_erf(u); U = gelu'(u) (both stored) */
  NBEST_EPI_BIAS_DROP_RES = 3, /* C = drop(acc + bias[n]) + R[m,n]                           */
  NBEST_EPI_DGELU = 4,         /* C = acc * U[m,n]   (U = gelu'(u) saved by BIAS_GELU, see U) */
  NBEST_EPI_RES = 5,           /* C = acc + R[m,n]                                           */
  NBEST_EPI_F32_SPLITK = 6     /* Cf32[N x ...] = sum over K-splits (weight gradient), fp32  */
};
typedef struct nbest_gemm_args {
  const void* A;
  const void* B;
  void* C;            /* dtype output (or fp32 when epilogue == NBEST_EPI_F32_SPLITK)          */
  const float* bias;  /* [N] or NULL                                                          */
  const void* R;      /* residual [M][ldr], dtype                                             */
  void* U;            /* GELU derivative gelu'(u) [M][ldu]: written by BIAS_GELU, read by DGELU.  NBEST_F32: float.
                         NBEST_BF16: ONE BYTE per element, fixed point q = round(200 g') + 26 (step 1/200 over
                         [-0.13, 1.145], 0 and 1 exact), ldu in bytes                                       */
  void* ws;           /* split-K slabs: >= nbest_gemm_ws_bytes(args)                           */
  size_t ws_bytes;
  int64_t M, N, K;
  int64_t lda, ldb, ldc, ldr, ldu;
  int32_t trans_a, trans_b;
  int32_t epilogue;
  int32_t dtype;
  int32_t accumulate; /* F32_SPLITK only: C += result instead of C = result                    */
  float drop_p;
  uint32_t drop_stream;
  uint64_t seed;
  float* colsum_out;        /* optional [N]: column sums of the OUTPUT C (fp32, before rounding) = the bias
                               gradient of the layer that produced A; fused into the epilogue (not for
                               F32_SPLITK).  Needs ws >= nbest_gemm_ws_bytes().                            */
  int32_t colsum_accumulate; /* colsum_out += instead of = */
  int32_t flags;             /* NBEST_GEMM_DEFER_REDUCE: F32_SPLITK leaves the per-split partial slabs in ws
                                (layout [splits][M][N] fp32) and does NOT launch the reduce; C is untouched */
  const void* B_packed;      /* optional (bf16, trans_a = trans_b = 0): the same matrix as B, pre-packed by nbest_pack_weights for tiles of
                                b_pack_bn columns.  Used when the kernel chosen for this shape has that tile width (the LDS-DMA of the B
                                tile becomes a linear copy); otherwise B is read.  Results are identical either way.               */
  int32_t b_pack_bn;         /* 256 | 192 (nbest_pack_bn(N)), 0 = none */
  int32_t pad_;
} nbest_gemm_args;
#define NBEST_GEMM_DEFER_REDUCE 1
size_t nbest_gemm_ws_bytes(const nbest_gemm_args* a);
int nbest_gemm(const nbest_gemm_args* a, nbest_stream_t stream);

/* Two weight gradients dW = dY^T . X (bf16 operands, trans_a = trans_b = 1, NBEST_EPI_F32_SPLITK) that share the token dimension K
 * and the column count N, in ONE launch: the 256 x 256 output tiles of `b` are appended to those of `a`, every tile takes the same
 * K-splits and one reduce launch writes both gradients (a->C and b->C; a->ws / a->ws_bytes hold the slabs of both:
 * >= nbest_wgrad_pair_ws_bytes).  Used by nbest_encoder_backward for the QKV ([3H, H]) and attention-output ([H, H]) gradients of
 * a layer - the backward of the two nn.Linear of installed modeling_bert.py:154-177 / 282-293 - whose tile counts (27 + 9) add up to
 * the FFN gradients' (36): the small gradient no longer runs as a launch of 9 tiles x 28 splits of its own.  Results equal
 * nbest_gemm's on each problem up to the fp32 summation order over K-splits.  M1, M2, N multiples of 256, same `accumulate`;
 * nbest_wgrad_pair_ws_bytes returns 0 and nbest_wgrad_pair NBEST_ERR_SHAPE for pairs that do not fit (issue two nbest_gemm).   */
size_t nbest_wgrad_pair_ws_bytes(const nbest_gemm_args* a, const nbest_gemm_args* b);
int nbest_wgrad_pair(const nbest_gemm_args* a, const nbest_gemm_args* b, nbest_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fp8 forward GEMMs (dtype path NBEST "fp8w": BASELINE configs[4], "fp8 weights (CDNA4 fp8 MFMA)")
 * C[M][N] (bf16) = epi((A8[M][K] . W8[N][K]^T) * out_scale + bias): both operands OCP e4m3 (one byte per element, k-contiguous),
 * on the block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales; out_scale = 1 / (per-matrix weight scale).
 * Replaces the same nn.Linear forwards as nbest_gemm (installed modeling_bert.py:154-177, 282-293, 325-351); the
 * weight gradients have their own entry point (nbest_wgrad_fp8 below).  Forward epilogues: NBEST_EPI_BIAS, NBEST_EPI_BIAS_GELU
 * (writes C = gelu bf16, U = gelu' 8-bit, C8 = e4m3 copy of gelu for the next GEMM), NBEST_EPI_BIAS_DROP_RES; dgrad epilogues
 * (B = the TRANSPOSED e4m3 weight copy, A = e4m3 gradient copy): NBEST_EPI_NONE, NBEST_EPI_RES, NBEST_EPI_DGELU.
 * N % 256 == 0, K % 64 == 0.  C may be NULL for BIAS_GELU / DGELU when C8 is given (every reader takes the e4m3 copy).          */
typedef struct nbest_gemm_fp8_args {
  const void* A;      /* e4m3 [M][lda] */
  const void* B;      /* e4m3 [N][ldb] (the weight matrix as stored, [out][in]) */
  void* C;            /* bf16 [M][ldc] */
  const float* bias;  /* [N] */
  const void* R;      /* bf16 residual [M][ldr] (BIAS_DROP_RES) */
  void* U;            /* 8-bit gelu' [M][ldu] (BIAS_GELU), format of nbest_gemm_args::U */
  void* C8;           /* e4m3 copy of C [M][ldc8] (BIAS_GELU) */
  int64_t M, N, K;
  int64_t lda, ldb, ldc, ldr, ldu, ldc8;
  int32_t epilogue;
  float out_scale;
  float drop_p;
  uint32_t drop_stream;
  uint64_t seed;
  const float* out_scale_dev; /* optional DEVICE scalar that overrides out_scale (scales produced on the device by
                                 nbest_quantize_weights_fp8: no host round trip) */
  /* ---- scaled A operand (every epilogue): A8 = e4m3(a * s) with s = 2^floor(log2(T / amax)) of the amax stored (as float
   * bits) at a_amax - the DELAYED per-tensor scale its producer used: a gradient tensor (dgrad epilogues NONE, RES, DGELU:
   * T = 56, 8 x headroom) or, round 4, a forward activation (BIAS, BIAS_GELU, BIAS_DROP_RES: T = 224, 2 x headroom); the
   * accumulator is divided by s.  NULL = unit scale.
   * DGELU: C = acc * gelu'(U) in bf16; optional C8 = e4m3(C * s_c), s_c from *c8_amax_prev; max(|C|) recorded into the slot block c8_amax_new
   * (NBEST_AMAX_TENSOR_WORDS words, see nbest_fp8_amax_fold);
   * optional colsum_out[N] (+)= column sums of C (the FFN-up bias gradient), needs ws >= nbest_gemm_fp8_ws_bytes().
   * BIAS_GELU: C8 = e4m3(gelu * s_c) with the same two fields (the activation scale of the FFN-down GEMM's input).          */
  const uint32_t* a_amax;
  const uint32_t* c8_amax_prev;
  uint32_t* c8_amax_new;
  float* colsum_out;
  int32_t colsum_accumulate;
  int32_t pad;
  void* ws;
  size_t ws_bytes;
  const void* B_packed; /* optional: B pre-packed by nbest_pack_weights_fp8 for tiles of b_pack_bn columns (256 | 128); used when the kernel
                           chosen for the shape has that tile width, otherwise B is read.  Identical results.                        */
  int32_t b_pack_bn;
  int32_t pad3;
} nbest_gemm_fp8_args;
size_t nbest_gemm_fp8_ws_bytes(const nbest_gemm_fp8_args* a);
int nbest_gemm_fp8(const nbest_gemm_fp8_args* a, nbest_stream_t stream);
/* fp8 weight gradient: dW[M][N] (fp32) (+)= sum over K tokens of dY8[k][m] * X8[k][n] / s, both operands TOKEN-major e4m3
 * ([K][lda] / [K][ldb], as their producers wrote them: the reduction dimension is read through transposed LDS reads),
 * s = the gradient scale of dY8 (from the float bits at a_amax, NULL = 1) times the activation scale of X8 (x_amax, NULL = 1).
 * M, N multiples of 256; split-K over tokens into ws (>= nbest_wgrad_fp8_ws_bytes) + deterministic reduce.  Replaces the
 * weight-gradient half of the nn.Linear backward. */
size_t nbest_wgrad_fp8_ws_bytes(int64_t M, int64_t N, int64_t K);
int nbest_wgrad_fp8(const void* dY8, const void* X8, float* dW, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                    int64_t ldc, const uint32_t* a_amax, const uint32_t* x_amax, int accumulate, void* ws, size_t ws_bytes,
                    nbest_stream_t stream);
/* Two fp8 weight gradients with the same token dimension K and column count N in ONE launch (the e4m3 counterpart of nbest_wgrad_pair;
 * nbest_encoder_backward pairs the Q|K|V and attention-output gradients of a layer): problem b's 256 x 256 output tiles are appended below
 * problem a's, each keeps its own gradient scale (amax_a / amax_b), one reduce writes dWa and dWb.  Ma, Mb, N multiples of 256; ws >=
 * nbest_wgrad_fp8_pair_ws_bytes (0 = the pair does not fit: issue two nbest_wgrad_fp8).                                                */
size_t nbest_wgrad_fp8_pair_ws_bytes(int64_t Ma, int64_t Mb, int64_t N, int64_t K);
int nbest_wgrad_fp8_pair(const void* dY8a, const void* X8a, float* dWa, int64_t Ma, int64_t lda_a, int64_t ldb_a, int64_t ldc_a,
                         const uint32_t* amax_a, const uint32_t* xamax_a, const void* dY8b, const void* X8b, float* dWb, int64_t Mb,
                         int64_t lda_b, int64_t ldb_b, int64_t ldc_b, const uint32_t* amax_b, const uint32_t* xamax_b, int64_t N, int64_t K,
                         int accumulate, void* ws, size_t ws_bytes, nbest_stream_t stream);
/* bf16 [n] -> e4m3 [n], unit scale, saturating at +-448 (activations that feed an fp8 GEMM); n % 8 == 0 */
int nbest_cast_bf16_to_fp8(const void* src, void* dst, int64_t n, nbest_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  scaled-dot-product attention with a per-sample key-padding mask
 * replaces eager_attention_forward (installed modeling_bert.py:111-136); mask rule of
 * /root/reference/models/model.py:43 is applied by the CALLER (key_mask = ids > 0, uint8 [B,S]).
 *   qkv [M][3H] (Q | K | V, head h = columns h*d .. h*d+d of each third), ctx [M][H],
 *   lse [B][heads][S] = log-sum-exp of the scaled masked scores (saved for backward).
 *   P = softmax(Q K^T / sqrt(d) + (mask ? 0 : -inf)) ; ctx = drop(P) V.  d must be 64; S <= 512
 *   (bf16: S <= 256).                                                                              */
int nbest_attention_fwd(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S,
                        int heads, int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream,
                        nbest_stream_t stream);
/* dqkv [M][3H] receives dQ | dK | dV (overwritten, no accumulation).  dbias != NULL: dbias[3H] (+)= column
 * sums of dqkv (the Q|K|V bias gradient), fused into the kernel; ws >= nbest_attention_bwd_ws_bytes().   */
size_t nbest_attention_bwd_ws_bytes(int B, int S, int heads);
int nbest_attention_bwd(const void* qkv, const uint8_t* key_mask, const void* ctx, const void* dctx,
                        const float* lse, void* dqkv, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                        int B, int S, int heads, int d, int dtype, float drop_p, uint64_t seed,
                        uint32_t drop_stream, nbest_stream_t stream);

/* The same pair with the dropout decisions handed from the forward to the backward (bf16, S <= 256): the forward writes its
 * keep decisions as bit words keep[B * heads][ceil(S/32)][32 ceil(S/32)] (bit j of word (key block kb, query q) = key 32 kb + j
 * kept; nbest_attention_keep_bytes() bytes, 0 when the shape has no such path), the backward reads them instead of hashing
 * every score element again.  Same decisions, hence the same results as the plain pair up to fma contraction in the last bit;
 * keep may be NULL (then identical to the plain pair). */
size_t nbest_attention_keep_bytes(int B, int S, int heads);
int nbest_attention_fwd_keep(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S,
                             int heads, int d, int dtype, float drop_p, uint64_t seed, uint32_t drop_stream,
                             void* keep, nbest_stream_t stream);
int nbest_attention_bwd_keep(const void* qkv, const uint8_t* key_mask, const void* ctx, const void* dctx,
                             const float* lse, void* dqkv, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                             int B, int S, int heads, int d, int dtype, float drop_p, uint64_t seed,
                             uint32_t drop_stream, const void* keep, nbest_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5  LayerNorm over the hidden dimension (BertSelfOutput / BertOutput LayerNorm,
 * installed modeling_bert.py:282-293, 340-351).  stats[m] = {mean, rstd} fp32.                      */
int nbest_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                        int64_t M, int H, float eps, int dtype, nbest_stream_t stream);
/* dx = LN'(dy); dgamma = sum_m dy*xhat; dbeta = sum_m dy (+= when accumulate).
 * The LayerNorm input was x = drop(dense_out) + residual, so two gradients leave this kernel:
 *   dx      - gradient of the residual branch (unmasked), and
 *   dx_drop - gradient of dense_out = dropout mask (regenerated from seed/drop_stream) applied to dx;
 *             only written when drop_p > 0 (otherwise the caller uses dx for both; dx_drop may be NULL).
 * dbias != NULL: dbias = sum_m dx_drop (gradient of the dense layer's bias).
 * ws: >= nbest_rowred_ws_bytes(M, H).                                                               */
int nbest_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                        void* dx_drop, float* dgamma, float* dbeta, float* dbias, int64_t M, int H, int dtype,
                        int accumulate, float drop_p, uint64_t seed, uint32_t drop_stream, void* ws,
                        size_t ws_bytes, nbest_stream_t stream);
size_t nbest_rowred_ws_bytes(int64_t M, int64_t N);
/* out[n] (+)= sum_m X[m][n]   (bias gradients).  ws: >= nbest_rowred_ws_bytes(M, N).               */
int nbest_colsum(const void* X, float* out, int64_t M, int64_t N, int64_t ld, int dtype, int accumulate,
                 void* ws, size_t ws_bytes, nbest_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7  STC heads + losses, forward and analytic backward in one call
 * replaces HierarchicalClassifier.forward (/root/reference/models/modules/hierarchical_classifier.py:35-60),
 * cal_total_loss / cal_ce_loss (/root/reference/n_best_asr_bert.py:145-195), convert_labels and
 * onehot_to_scalar (/root/reference/utils/STC_util.py:4-7,29-51).
 * Label space: n_top top labels; bottom_of[...] lists, for top t, its bottom ids
 * bottom_ids[bottom_off[t] .. bottom_off[t+1]); tops with >= 2 bottoms own a softmax head whose rows
 * in the concatenated head matrix Wh [R][H] start at head_row[t] (single-bottom tops: -1); rows
 * 0..n_top-1 of Wh are the top (sigmoid) classifier.  R = n_top + sum of multi-value head sizes.
 *   cls     = hidden[b * cls_stride .. +H]  (raw CLS row, models/model.py:46-47), dtype
 *   top     [B][n_top], bott [B][R - n_top] (softmax heads concatenated in head order), final [B][n_bottom]
 *   loss_parts[4] = {BCE_sum(final,y), BCE_sum(top, y.B2T), mean_k NLL_sum_k, 0}  (fp32, overwritten)
 *   dcls [B][H] fp32 = d(sum of the three)/d cls ; dWh [R][H], dbh [R] fp32 (overwritten unless accumulate)
 * feature dropout: an independent mask per linear layer (hierarchical_classifier.py:41,46).
 * ws: >= nbest_heads_ws_bytes(B, R, H).                                                             */
typedef struct nbest_label_space {
  int32_t n_top, n_bottom, n_rows;
  const int32_t* bottom_off; /* [n_top+1] device */
  const int32_t* bottom_ids; /* [n_bottom] device */
  const int32_t* head_row;   /* [n_top] device   */
} nbest_label_space;
size_t nbest_heads_ws_bytes(int B, int R, int H);
int nbest_stc_heads(const void* hidden, int64_t cls_stride, const float* Wh, const float* bh,
                    const nbest_label_space* ls, const float* labels, float* top, float* bott, float* final_scores,
                    float* loss_parts, float* dcls, float* dWh, float* dbh, int B, int H, int dtype,
                    int need_grad, int accumulate, float drop_p, uint64_t seed, uint32_t drop_stream, void* ws,
                    size_t ws_bytes, nbest_stream_t stream);
/* Backward of the heads for ARBITRARY upstream gradients dtop [B][n_top], dbott [B][R - n_top], dfin [B][n_bottom] (fp32) - what
 * torch autograd hands to the heads when the reference's loop calls total_loss.backward() on a loss it built itself
 * (/root/reference/n_best_asr_bert.py:255-264; nbest_amd.model's autograd bridge).  `ws` must be the workspace of the
 * nbest_stc_heads call (need_grad = 0 is enough) that produced top / bott for the same inputs and dropout seed: it holds the fp32
 * CLS rows and the dropout bits.  dcls [B][H] overwritten; dWh / dbh overwritten unless accumulate.                                */
int nbest_stc_heads_vjp(const float* Wh, const nbest_label_space* ls, const float* top, const float* bott, const float* dtop,
                        const float* dbott, const float* dfin, float* dcls, float* dWh, float* dbh, int B, int H, int accumulate,
                        float drop_p, uint64_t seed, uint32_t drop_stream, void* ws, size_t ws_bytes, nbest_stream_t stream);

/* K8  pooled-CLS MSE auxiliary loss (--add_l2_loss): nn.MSELoss() between the ASR and transcript CLS
 * rows, /root/reference/n_best_asr_bert.py:166-170,574.  loss[0] = mean((a-t)^2);
 * da += grad_scale * 2(a-t)/(B*H); dt = -grad_scale * 2(a-t)/(B*H)  (the transcript branch is NOT detached). */
int nbest_cls_mse(const void* hidden_a, int64_t stride_a, const void* hidden_t, int64_t stride_t, float* loss,
                  float* da, float* dt, int B, int H, int dtype, float grad_scale, nbest_stream_t stream);

/* scatter a [B][H] fp32 CLS gradient into row 0 of every sequence of dhidden [B*S][H] (all other
 * rows zero): the backward of sequence_output[:, 0, :].                                             */
int nbest_cls_grad_scatter(const float* dcls, void* dhidden, int B, int S, int H, int dtype,
                           nbest_stream_t stream);

/* K10 device decode of pred_one_sample (/root/reference/n_best_asr_bert.py:198-215):
 * pred[b][t] = predicted bottom label index when top[b][t] > 0.5 (strict), else -1; for a
 * multi-value top the argmax of its head (first maximum), -1 when none_flag[that bottom] != 0.      */
int nbest_stc_decode(const float* top, const float* bott, const nbest_label_space* ls,
                     const uint8_t* none_flag, int32_t* pred, int B, nbest_stream_t stream);
/* `pred` may be any device-ACCESSIBLE memory - in particular mapped pinned host memory (hipHostMalloc): the rows then land on the
 * host without a copy command.  nbest_stream_stamp writes `value` to *flag (device-accessible, system-scope release) in stream order:
 * stamped after a decode into host memory it tells a polling host thread that the rows are complete - the per-step prediction
 * hand-off of the reference's loop (n_best_asr_bert.py:283-288) without a hipMemcpy, an event or a stream synchronisation. */
int nbest_stream_stamp(int32_t* flag, int32_t value, nbest_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K9  multi-tensor BertAdam over flat arenas
 * replaces BertAdam.step (/root/reference/models/optimization.py:237-302) with the per-parameter
 * grouping of /root/reference/n_best_asr_bert.py:540-561: per tensor g *= min(1, 1/(||g||+1e-6));
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; u = m/(sqrt(v)+eps) [+ wd p]; p -= lr*lr_mult * u,
 * lr_mult = warmup-linear schedule value for this step (optimization.py:162-171; one scalar: every
 * tensor that receives gradients shares the same step count).
 * The arenas p, g, m, v are fp32; tensors are described by device descriptors ordered by offset.
 * block_start[t] = sum_{u<t} ceil(numel[u] / nbest_bertadam_chunk()); n_blocks = that sum over all
 * tensors.  If p_lowp != NULL the updated parameters are also written as bf16 at the same element
 * offsets (the compute copy the GEMMs read).  ws: >= (n_blocks + n_tensors) floats.                 */
typedef struct nbest_tensor_desc {
  int64_t offset;
  int64_t numel;
  float lr;            /* base learning rate of the tensor (bert_lr / lr) */
  float wd;            /* 0.01, or 0 for bias / LayerNorm */
  int32_t active;      /* 0: no gradient this step (skipped, like `p.grad is None`) */
  int32_t block_start;
} nbest_tensor_desc;
int nbest_bertadam_chunk(void);
int nbest_bertadam_step(float* p, float* g, float* m, float* v, void* p_lowp, const nbest_tensor_desc* descs,
                        int n_tensors, int n_blocks, float lr_mult, float b1, float b2, float eps,
                        float max_grad_norm, void* ws, size_t ws_bytes, nbest_stream_t stream);
/* The same step in two halves over a range of blocks [blk_lo, blk_hi) (new functionality: the optimizer sharded over data-parallel
 * ranks).  norms: partial[blk] = sum of squares of block blk's gradient elements, for the blocks of the range (the caller zeroes
 * `partial` [n_blocks] and SUM-all-reduces it over the ranks).  update: clip coefficients of all tensors from `partial`
 * (coef [n_tensors], scratch), then BertAdam on the blocks of the range.  norms + update over [0, n_blocks) == nbest_bertadam_step. */
int nbest_bertadam_norms(const float* g, const nbest_tensor_desc* descs, int n_tensors, int n_blocks, int blk_lo, int blk_hi,
                         float* partial, nbest_stream_t stream);
int nbest_bertadam_update(float* p, const float* g, float* m, float* v, void* p_lowp, const nbest_tensor_desc* descs,
                          int n_tensors, int n_blocks, int blk_lo, int blk_hi, const float* partial, float* coef,
                          float lr_mult, float b1, float b2, float eps, float max_grad_norm, nbest_stream_t stream);
/* Transposed bf16 copy of the weight matrices (same element offsets in `dst` as in `src`): matrix t is
 * [rows][cols] in src and [cols][rows] in dst.  The backward's dgrad GEMMs read this copy so that both of
 * their operands are k-contiguous (no transposed LDS reads).  descs: DEVICE array ordered by tile_start,
 * tile_start[t] = sum_{u<t} ceil(rows/64)*ceil(cols/64); n_tiles = that sum over all matrices.          */
typedef struct nbest_matrix_desc {
  int64_t offset;
  int32_t rows, cols;
  int32_t tile_start, pad;
} nbest_matrix_desc;
/* Recording side of the delayed fp8 scales.  Every kernel that records a tensor's amax (gamax_new / aamax_new of the encoder
 * descriptor, c8_amax_new of nbest_gemm_fp8_args) is handed a block of NBEST_AMAX_TENSOR_WORDS uint32 words for that tensor, not
 * one word: it raises ONE of 16 words spaced 256 bytes apart (picked by its block index), because device-scope accesses to a
 * single word serialise (about 1 ns each: the 65 536 waves of an e4m3 cast spent 100 us on them).  nbest_fp8_amax_fold writes
 * out[t] = max over the 16 words of tensor t (the single word the NEXT pass reads as *_amax_prev / a_amax) and zeroes the slots.
 * slots: uint32 [n_tensors][NBEST_AMAX_TENSOR_WORDS], zero-initialised by the caller once. */
#define NBEST_AMAX_TENSOR_WORDS 1024
int nbest_fp8_amax_fold(void* slots, void* out, int32_t n_tensors, nbest_stream_t stream);
/* e4m3 copy of the weight matrices (same element offsets, ONE byte per element) from the fp32 master, one scale per
 * matrix: w8 = e4m3(w * 2^floor(log2(224 / max|w|))); inv_scale[i] = 1 / scale of matrix i (device, [n_matrices]).
 * w8t (optional): the same e4m3 values written transposed ([cols][rows] at the matrix' offset; n_tiles = 64x64 tiles over all
 * matrices, descs[].tile_start as for nbest_transpose_weights) - the B operand of the fp8 dgrad GEMMs.
 * ws: >= 4 * n_matrices bytes.  Run after every optimizer step, like nbest_transpose_weights.                       */
int nbest_quantize_weights_fp8(const float* master, void* w8, void* w8t, const nbest_matrix_desc* descs, int n_matrices,
                               int n_tiles, float* inv_scale, void* ws, size_t ws_bytes, nbest_stream_t stream);
int nbest_transpose_weights(const void* src, void* dst, const nbest_matrix_desc* descs, int n_matrices, int n_tiles,
                            nbest_stream_t stream);
/* Weight matrices pre-packed for the k-contiguous GEMM kernels (nbest_gemm_args::B_packed).  For each matrix of `descs` (offset in
 * elements into both arenas, rows = N, cols = K, pad = tile width nbest_pack_bn(N) > 0, cols % 32 == 0, rows % pad == 0; tile_start =
 * running sum of (rows / pad) * (cols / 32) = index of the matrix' first (tile column, K stage) block; n_stages = that sum over all
 * matrices) `dst` receives, for every tile column and every 32-deep K stage, the pad x 32 tile in the exact order the kernel's LDS image
 * has (16-byte chunk swizzle and row permutation applied).  Same bytes as the source, N * K elements per matrix at the same offset.
 * Once per optimizer step, beside nbest_transpose_weights / the bf16 copy refresh of nbest_bertadam_step (the nn.Linear weights of the
 * installed modeling_bert.py:154-177, 282-293, 325-351 do not change inside a step).                                                   */
int nbest_pack_bn(int64_t N);
/* the same for the e4m3 copies read by nbest_gemm_fp8 (descs[].pad = nbest_pack_bn_fp8(rows, cols) = 256 | 128, cols % 64 == 0,
 * tile_start / n_stages in units of (tile column, 64-byte K stage) blocks; one byte per element at the arena's element offsets)      */
int nbest_pack_bn_fp8(int64_t N, int64_t K);
int nbest_pack_weights_fp8(const void* src, void* dst, const nbest_matrix_desc* descs, int n_matrices, int n_stages, nbest_stream_t stream);
int nbest_pack_weights(const void* src, void* dst, const nbest_matrix_desc* descs, int n_matrices, int n_stages, nbest_stream_t stream);
/* fp32 -> bf16 copy of an arena (initial compute copy / after loading a checkpoint) */
int nbest_cast_f32_to_bf16(const float* src, void* dst, int64_t n, nbest_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The whole encoder stack in one call (launch-overhead-free path used by training):
 * embeddings -> L x {QKV GEMM, attention, out-proj+residual, LN, FFN-up+GELU, FFN-down+residual, LN}.
 * Replaces the encoder(...) calls at /root/reference/models/model.py:43-45,54-56 and autograd
 * through them (/root/reference/n_best_asr_bert.py:264).
 * Parameter arenas: `wts` = matrices/tables in `dtype`, `prm` = the fp32 master arena (biases,
 * LayerNorm), `grad` = fp32 gradient arena; all three share ONE element-offset table
 * (nbest_encoder_layout).  act: activation stash written by forward, consumed by backward,
 * >= nbest_encoder_act_bytes(); ws: scratch >= nbest_encoder_ws_bytes().                            */
typedef struct nbest_layer_offsets {
  int64_t wqkv, bqkv, wo, bo, ln1_g, ln1_b, w1, b1, w2, b2, ln2_g, ln2_b;
} nbest_layer_offsets;
typedef struct nbest_encoder_desc {
  int32_t dtype, B, S, H, L, heads, F;
  int32_t vocab, max_pos, n_types;
  float ln_eps, hidden_drop, attn_drop;
  int64_t word_pad_id, pos_pad_id; /* padding_idx rows (no gradient), -1 = none */
  int64_t off_word, off_pos, off_type, off_emb_ln_g, off_emb_ln_b;
  const nbest_layer_offsets* layers_host; /* [L], HOST memory */
  uint64_t seed;
  uint32_t drop_stream_base; /* distinct per encoder pass within a step */
  int32_t wgrad_events_n;    /* entries of wgrad_events (0 = no timing) */
  /* optional in-step timing of the weight-gradient GEMMs (bench.py's roofline object): caller-created hipEvent_t
   * handles; nbest_encoder_backward records wgrad_events[2*i] before and [2*i+1] after the i-th weight-gradient
   * launch it enqueues (nbest_encoder_wgrad_launches_per_layer() per layer, highest layer first: FFN-down, FFN-up,
   * attention-out, QKV - or, bf16, FFN-down, FFN-up, QKV + attention-out as one nbest_wgrad_pair launch), on the
   * caller's stream, while i < wgrad_events_n / 2.  The library never creates, waits on or destroys events.      */
  void** wgrad_events;
  /* optional fp8 forward ("fp8w"; dtype must be NBEST_BF16): e4m3 copy of the weight arena (one byte per element at the
   * same element offsets) and its per-matrix inverse scales [4 L] (QKV, attention-out, FFN-up, FFN-down per layer), both
   * from nbest_quantize_weights_fp8.  The forward GEMMs then run on the block-scaled fp8 MFMA; their activation operands
   * are e4m3 copies written by the producers with a DELAYED per-tensor scale (aamax_* below); everything else is the bf16 path. */
  const void* w8;
  const float* w8_inv_scale;
  /* optional fp8 dgrads (needs w8): transposed e4m3 weight copy and the per-(layer, tensor) gradient amax history gamax_prev, uint32
   * float bits [4 L]: index 4 l + {0: FFN-down input gradient, 1: FFN-up input gradient (after GELU'), 2: attention-out
   * input gradient, 3: dQ|dK|dV}.  The backward always RECORDS this pass's amax (when gamax_new is non-NULL) into gamax_new, which is
   * a SLOT ARRAY uint32 [4 L][NBEST_AMAX_TENSOR_WORDS] (see nbest_fp8_amax_fold); with fp8_bwd != 0 the producers also write e4m3
   * copies scaled from gamax_prev and the four dgrad GEMMs of a layer run on the fp8 MFMA.  Between passes the caller folds the
   * slots into the history (nbest_fp8_amax_fold(gamax_new, gamax_prev, 4 L)) and sets fp8_bwd only once a history exists.   */
  const void* w8t;
  const uint32_t* gamax_prev;
  uint32_t* gamax_new;
  int32_t fp8_bwd;
  int32_t pad2;
  /* optional (bf16): the weight arena packed by nbest_pack_weights for the forward GEMMs (wpk: matrices as stored, [out][in]) and for
   * the dgrad GEMMs (wpkt: packed from the TRANSPOSED copy wts_t), at the arena's element offsets; NULL = the GEMMs read wts / wts_t */
  const void* wpk;
  const void* wpkt;
  /* optional (fp8w): w8 / w8t packed by nbest_pack_weights_fp8, at the arena's element offsets (bytes) */
  const void* w8p;
  const void* w8tp;
  /* REQUIRED by nbest_encoder_backward(with_embeddings): token indices of THIS pass sorted by word id (stable), int32 [B*S], device
   * memory - see nbest_embed_ln_bwd.  Set per call like `seed` (it belongs to the batch, not to the shape).                        */
  const int32_t* word_perm;
  /* fp8 forward: activation amax history aamax_prev, uint32 float bits [4 L]: index 4 l + {0: layer input x (QKV GEMM), 1: ctx
   * (attention-out GEMM), 2: x1 (FFN-up GEMM), 3: gelu(u) (FFN-down GEMM)}.  Every forward pass RECORDS this pass's amax into
   * aamax_new (when non-NULL), a SLOT ARRAY uint32 [4 L][NBEST_AMAX_TENSOR_WORDS] like gamax_new.  With fp8_act != 0 (a history
   * exists in aamax_prev) the producers write e4m3(a * s), s = 2^floor(log2(224 / amax_prev)), the GEMMs divide by s (and the fp8
   * weight gradients of the backward, which read the same copies, too); with fp8_act == 0 the forward is a CALIBRATION pass: it
   * runs the bf16 GEMMs and only records the amax.  After every step (after the backward, which reads the copies scaled by prev)
   * the caller folds: nbest_fp8_amax_fold(aamax_new, aamax_prev, 4 L), and sets fp8_act once a history exists.                  */
  const uint32_t* aamax_prev;
  uint32_t* aamax_new;
  int32_t fp8_act;
  int32_t pad4;
} nbest_encoder_desc;
size_t nbest_encoder_act_bytes(const nbest_encoder_desc* d);
size_t nbest_encoder_ws_bytes(const nbest_encoder_desc* d);
/* weight-gradient launches nbest_encoder_backward enqueues per layer for this descriptor: 3 when the attention-output gradient
 * rides with the QKV gradient (nbest_wgrad_pair; bf16, shapes that fit), else 4 */
int nbest_encoder_wgrad_launches_per_layer(const nbest_encoder_desc* d);
/* hidden_out: pointer to the final hidden states [M][H] inside act (returned through *hidden_out) */
int nbest_encoder_forward(const nbest_encoder_desc* d, const void* wts, const float* prm, const int64_t* ids,
                          const int64_t* seg, const int64_t* pos, const uint8_t* key_mask, void* act,
                          size_t act_bytes, void* ws, size_t ws_bytes, void** hidden_out, nbest_stream_t stream);
/* dhidden [M][H] (dtype) is consumed (overwritten): it always holds the running gradient w.r.t. the
 * input of the lowest layer processed so far.  Parameter gradients are written into `grad`
 * (overwritten; accumulate != 0: added).  Layers [layer_begin, layer_end) are processed in reverse;
 * with_embeddings != 0 also runs the embedding backward (needs layer_begin == 0).  Calling it in
 * chunks (L..k, k..0+embeddings) lets the host start the gradient all-reduce of finished layers
 * while the remaining backward runs.                                                                */
/* wts_t: optional arena holding the TRANSPOSED weight matrices (nbest_transpose_weights); NULL -> the dgrad
 * GEMMs read `wts` with transposed LDS reads.                                                            */
int nbest_encoder_backward(const nbest_encoder_desc* d, const void* wts, const void* wts_t, const float* prm, float* grad,
                           const int64_t* ids, const int64_t* seg, const int64_t* pos, const uint8_t* key_mask,
                           void* act, size_t act_bytes, void* dhidden, void* ws, size_t ws_bytes, int accumulate,
                           int layer_begin, int layer_end, int with_embeddings, nbest_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NBEST_HIP_H */
