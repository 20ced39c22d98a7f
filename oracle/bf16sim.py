"""Oracle, bf16-storage leg: the SAME fp32 CPU restatement (oracle/encoder.py, oracle/model.py), with values rounded to
bfloat16 at exactly the places where the HIP path keeps a bf16 tensor in HBM or feeds a bf16 MFMA operand; all
contractions, LayerNorm statistics, softmax, GELU and the heads stay fp32, like the fp32 accumulators of the kernels.

Purpose: |bf16-leg - fp32 reference| is the noise floor any bf16-storage implementation of this path shows on a given
case.  tests/golden/make_golden.py commits that floor per compared quantity; the GPU parity test bounds
|HIP bf16 - fp32 reference| by a small multiple of it instead of a hand-set tolerance.

Rounding points (forward; the gradient of each stored tensor is rounded too, as the backward kernels store bf16):
  weight matrices and the three embedding tables (compute copy `w16`; biases / LayerNorm parameters stay fp32),
  X0 = LN(embeddings), per layer qkv, softmax probabilities (MFMA operand of P.V), ctx, r1 = x + dense(ctx),
  x1 = LN(r1), hact = gelu(u) (gelu'(u), stored for the backward, is 8-bit fixed point), r2 = x1 + dense(hact), X = LN(r2).

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F

from .encoder import position_ids_for


def _r(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundAct(torch.autograd.Function):
    """stored activation: value rounded on the way forward, its gradient rounded on the way back"""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundWeight(torch.autograd.Function):
    """bf16 compute copy of an fp32 master weight: the weight gradient is produced and kept in fp32"""

    @staticmethod
    def forward(ctx, w):
        return _r(w)

    @staticmethod
    def backward(ctx, g):
        return g


def _q8(g):
    """gelu'(u) as the bf16 path keeps it: 8-bit fixed point, q = round(200 g' + 26) (half to even; step 1/200, 0 and 1 exact)"""
    return (torch.clamp(torch.round(g * 200.0 + 26.0), 0, 255) - 26.0) / 200.0


class _GeluStore(torch.autograd.Function):
    """FFN-up epilogue: hact = bf16(gelu(u)), stash q8(gelu'(u)); backward du = bf16(dh * gelu').  ``q8`` False: gelu' kept in
    bf16 instead - the plain bf16-storage implementation the 8-bit form is measured against (make_golden.py)."""

    @staticmethod
    def forward(ctx, u, q8=True):
        cdf = 0.5 * (1.0 + torch.erf(u * (1.0 / math.sqrt(2.0))))
        pdf = torch.exp(-0.5 * u * u) * (1.0 / math.sqrt(2.0 * math.pi))
        ctx.save_for_backward(_q8(cdf + u * pdf) if q8 else _r(cdf + u * pdf))
        return _r(u * cdf)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _r(g * d), None


class _AttnCore(torch.autograd.Function):
    """softmax(q k^T * scale + mask) v with the MFMA operand roundings of the attention kernels: P -> bf16 for P.V and
    dV, dS -> bf16 for dQ / dK; scores, softmax and dP in fp32; q, k, v, dO arrive already rounded (stored tensors)."""

    @staticmethod
    def forward(ctx, q, k, v, key_mask, scale):
        s = torch.matmul(q, k.transpose(-1, -2)) * scale
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(_r(p), v)
        ctx.save_for_backward(q, k, v, p, _r(o))
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, o = ctx.saved_tensors
        dv = torch.matmul(_r(p).transpose(-1, -2), do)
        dp = torch.matmul(do, v.transpose(-1, -2))
        delta = (do * o).sum(-1, keepdim=True)
        ds = _r(p * (dp - delta) * ctx.scale)
        dq = torch.matmul(ds, k)
        dk = torch.matmul(ds.transpose(-1, -2), q)
        return dq, dk, dv, None, None


def _e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def _fp8_scale(t, target=224.0):
    """per-tensor power-of-two scale of the fp8 copies: 2^floor(log2(target / max|t|)); weights map their maximum to 224
    (common.h fp8_scale_of), gradients to 56 = 8 x headroom under the one-step-old amax (fp8_gscale_of)"""
    a = t.abs().max()
    return 2.0 ** torch.floor(torch.log2(target / a)) if a > 0 else torch.tensor(1.0)


_ACT_SCALE = [True]     # False: unit-scale e4m3 activations (saturating at +-448) - the round-3 arithmetic, kept as a yardstick leg


def _e4m3_act(x):
    """e4m3 copy of a forward activation with its per-tensor power-of-two scale 2^floor(log2(224 / max|x|)) - 2 x headroom, as the
    weights (common.h fp8_ascale_of; the HIP path takes the amax from the previous step - delayed scaling; the parity tests run the
    same batch twice, so that is this tensor's own amax)"""
    if not _ACT_SCALE[0]:
        return _e4m3(x)
    s = _fp8_scale(x, 224.0)
    return _e4m3(x * s) / s


class _Fp8Linear(torch.autograd.Function):
    """"fp8w" GEMMs of one nn.Linear.  Forward: y = e4m3(x * s_x) / s_x . e4m3(W * s_w)^T / s_w + b (per-matrix scale s_w,
    nbest_quantize_weights_fp8; per-tensor activation scale s_x, see _e4m3_act).  Backward without ``bwd8``: the bf16 path's (bf16 gradient x bf16 activation
    -> fp32 weight gradient, bf16 dgrad on the bf16 weight copy).  With ``bwd8`` both are fp8: g8 = e4m3(g * s_g) / s_g feeds the
    dgrad g8 . e4m3(W * s_w) / s_w and the weight gradient g8^T . e4m3(x) (the forward's own e4m3 copy of the input), with
    the per-tensor gradient scale s_g (the HIP path takes it from the previous pass's amax of the same tensor; the parity test
    runs the same batch twice, so that is this tensor's own amax)."""

    @staticmethod
    def forward(ctx, x, w, b, bwd8, x_is_f32):
        s = _fp8_scale(w)
        w8 = _e4m3(w * s) / s
        x8 = _e4m3_act(x)
        ctx.save_for_backward(x8 if bwd8 else _r(x) if x_is_f32 else x, w8 if bwd8 else _r(w))
        ctx.bwd8, ctx.round_dx = bwd8, x_is_f32
        return F.linear(x8, w8) + b

    @staticmethod
    def backward(ctx, g):
        xs, wd = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        gw = g2
        if ctx.bwd8:
            sg = _fp8_scale(g2, 56.0)
            gw = _e4m3(g2 * sg) / sg
            dx = gw.view_as(g) @ wd
        else:
            dx = g @ wd
        if ctx.round_dx:
            dx = _r(dx)
        return dx, gw.t() @ xs.reshape(-1, xs.shape[-1]), g2.sum(0), None, None


class _GeluStoreFp8(torch.autograd.Function):
    """as _GeluStore; the fp8 forward's next GEMM reads the e4m3 copy of gelu(u) written by the same epilogue (from the
    fp32 value), so the rounding to bf16 is on the stored tensor only"""

    @staticmethod
    def forward(ctx, u):
        cdf = 0.5 * (1.0 + torch.erf(u * (1.0 / math.sqrt(2.0))))
        pdf = torch.exp(-0.5 * u * u) * (1.0 / math.sqrt(2.0 * math.pi))
        ctx.save_for_backward(_q8(cdf + u * pdf))
        return u * cdf

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _r(g * d)


ract, rw = _RoundAct.apply, _RoundWeight.apply


def _ln(x, mod):
    return F.layer_norm(x, (x.shape[-1],), mod.weight, mod.bias, mod.eps)


def encode(enc, ids, seg, fp8=False, fp8_bwd=False, q8=True):
    """oracle.encoder.OracleEncoder.forward with bf16 storage (dropout must be off: parity runs use p = 0); ``fp8``: the
    four forward GEMMs of every layer as the "fp8w" path runs them (e4m3 operands, the Q|K|V projection as ONE [3H, H]
    matrix with one scale), ``fp8_bwd``: their four dgrads in fp8 as well (see _Fp8Linear)"""
    cfg = enc.cfg
    key_mask = ids > 0                                                     # quirk Q1 (models/model.py:43)
    if seg is None:
        seg = torch.zeros_like(ids)
    pos = position_ids_for(cfg, ids)
    E = enc.embeddings
    pad_w = cfg.pad_token_id
    pad_p = cfg.pad_token_id if cfg.family in ("roberta", "xlm-roberta") else None
    e = (F.embedding(ids, rw(E.word_embeddings.weight), padding_idx=pad_w) + F.embedding(seg, rw(E.token_type_embeddings.weight))
         + F.embedding(pos, rw(E.position_embeddings.weight), padding_idx=pad_p))
    x = ract(_ln(e, E.LayerNorm))
    B, S, H = x.shape
    nh = cfg.num_attention_heads
    d = H // nh
    for lyr in enc.encoder.layer:
        a = lyr.attention.self
        split = lambda t: t.view(B, S, nh, d).transpose(1, 2)
        ao = lyr.attention.output
        if fp8:
            wqkv = torch.cat([a.query.weight, a.key.weight, a.value.weight])
            bqkv = torch.cat([a.query.bias, a.key.bias, a.value.bias])
            q, k, v = ract(_Fp8Linear.apply(x, wqkv, bqkv, fp8_bwd, False)).split(H, dim=-1)
        else:
            q, k, v = (ract(F.linear(x, rw(m.weight), m.bias)) for m in (a.query, a.key, a.value))
        o = _AttnCore.apply(split(q), split(k), split(v), key_mask, 1.0 / math.sqrt(d))
        ctx = ract(o.transpose(1, 2).reshape(B, S, H))
        if fp8:
            r1 = ract(_Fp8Linear.apply(ctx, ao.dense.weight, ao.dense.bias, fp8_bwd, False) + x)
            x1 = ract(_ln(r1, ao.LayerNorm))
            # gelu(u) reaches the next GEMM as e4m3 of the fp32 value; its bf16 copy feeds the weight gradient
            h32 = _GeluStoreFp8.apply(_Fp8Linear.apply(x1, lyr.intermediate.dense.weight, lyr.intermediate.dense.bias, fp8_bwd, False))
            r2 = ract(_Fp8Linear.apply(h32, lyr.output.dense.weight, lyr.output.dense.bias, fp8_bwd, True) + x1)
        else:
            r1 = ract(F.linear(ctx, rw(ao.dense.weight), ao.dense.bias) + x)
            x1 = ract(_ln(r1, ao.LayerNorm))
            hact = _GeluStore.apply(F.linear(x1, rw(lyr.intermediate.dense.weight), lyr.intermediate.dense.bias), q8)
            r2 = ract(F.linear(hact, rw(lyr.output.dense.weight), lyr.output.dense.bias) + x1)
        x = ract(_ln(r2, lyr.output.LayerNorm))
    return x[:, 0, :]


def forward(model, input_ids, trans_input_ids=None, seg_ids=None, trans_seg_ids=None, fp8=False, fp8_bwd=False, q8=True, act_scale=True):
    """oracle.model.OracleModel.forward (classifier_input_type 'asr') on the bf16-storage encoder; heads in fp32.
    ``act_scale`` False (with fp8): unit-scale e4m3 activations, what the fp8 forward did before it had activation scales"""
    _ACT_SCALE[0] = bool(act_scale)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            assert m.p == 0.0 or not model.training
    if model.family == "xlm-roberta":
        seg_ids = trans_seg_ids = None
    asr_cls = encode(model.bert_encoder, input_ids, seg_ids, fp8, fp8_bwd, q8)
    trans_cls = encode(model.bert_encoder, trans_input_ids, trans_seg_ids, fp8, fp8_bwd, q8) if trans_input_ids is not None else None
    top, bottoms, final = model.clf(asr_cls)
    return top, bottoms, final, asr_cls, trans_cls
