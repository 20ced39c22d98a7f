"""GPU parity of every hand-written HIP kernel (called through the C-ABI) against a plain PyTorch
fp32 restatement of the same op on identical seeded inputs.  Tolerances: fp32 path 1e-4 relative to
the tensor scale (north_star), bf16 path 1e-2 relative to the tensor scale."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import nbest_amd  # noqa: E402,F401
from nbest_amd import hipabi as hb  # noqa: E402

DEV = "cuda"
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kernel_parity.log")


def _log(msg):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(msg + "\n")


def close(name, got, ref, tol):
    got, ref = got.float(), ref.float()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item() / scale
    _log("%-58s rel_err=%.3e tol=%.1e scale=%.3e %s" % (name, err, tol, scale, "OK" if err <= tol else "FAIL"))
    assert math.isfinite(err) and err <= tol, "%s: rel err %.3e > %.1e (scale %.3e)" % (name, err, tol, scale)


def rnd(*shape, dtype=torch.float32, s=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * s).to(DEV).to(dtype)


def tol_of(dtype):
    return 1e-4 if dtype == torch.float32 else 1e-2


def gelu(u):
    return 0.5 * u * (1 + torch.erf(u / math.sqrt(2)))


def dgelu(u):
    return 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1)])
def test_gemm_layouts_and_epilogues(dtype, ta, tb):
    # ragged M for the activation-major GEMMs; ragged K (token dim) for the weight-gradient GEMM
    M, N, K = (200, 256, 192) if not ta else (256, 128, 1000)
    A = rnd(K, M, dtype=dtype, seed=1) if ta else rnd(M, K, dtype=dtype, seed=1)
    B = rnd(K, N, dtype=dtype, seed=2) if tb else rnd(N, K, dtype=dtype, seed=2)
    Af = (A.float().t() if ta else A.float())
    Bf = (B.float() if tb else B.float().t())
    ref = Af @ Bf
    tol = tol_of(dtype)
    tag = "gemm[%s ta=%d tb=%d]" % (str(dtype)[6:], ta, tb)
    close(tag + " none", hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_NONE), ref, tol)
    bias = rnd(N, seed=3)
    close(tag + " bias", hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_BIAS, bias=bias), ref + bias, tol)
    out, U = hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_BIAS_GELU, bias=bias)
    # gelu' travels as float in the fp32 path and as 8-bit fixed point (step 1/200, exact 0 and 1) in the bf16 path
    assert U.dtype == (torch.float32 if dtype == torch.float32 else torch.uint8)
    if dtype == torch.float32:
        close(tag + " bias_gelu.U (= gelu' of the pre-activation)", U, dgelu(ref + bias), tol)
    else:
        gerr = (hb.gelu_d_decode(U) - dgelu(ref + bias)).abs().max().item()
        _log("%-58s abs_err=%.3e (quantisation step 5e-3)" % (tag + " bias_gelu.U 8-bit", gerr))
        assert gerr <= 2.6e-3 + 2e-3, gerr          # half a step + the bf16-input GEMM error on the pre-activation
    close(tag + " bias_gelu.C", out, gelu(ref + bias), tol)
    R = rnd(M, N, dtype=dtype, seed=4)
    close(tag + " bias_res", hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_BIAS_DROP_RES, bias=bias, R=R), ref + bias + R.float(), tol)
    close(tag + " res", hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_RES, R=R), ref + R.float(), tol)
    Uf = dgelu(rnd(M, N, seed=5))                                   # a gelu' field: values in [-0.13, 1.13]
    Uin = Uf if dtype == torch.float32 else hb.gelu_d_encode(Uf)
    Ud = hb.gelu_d_decode(Uin)
    if dtype != torch.float32:
        assert (Ud - Uf).abs().max().item() <= 2.51e-3
        assert torch.equal(hb.gelu_d_decode(hb.gelu_d_encode(torch.tensor([0.0, 1.0], device=DEV))), torch.tensor([0.0, 1.0], device=DEV))
    cs = torch.zeros(N, device=DEV)
    close(tag + " dgelu", hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_DGELU, U=Uin, colsum_out=cs), ref * Ud, tol)
    close(tag + " dgelu fused column sums", cs, (ref * Ud).sum(0), 5 * tol)
    got = hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_F32_SPLITK)
    assert got.dtype == torch.float32
    close(tag + " f32_splitk", got, ref, tol)
    acc = got.clone()
    hb.gemm(A, B, M, N, K, ta, tb, hb.EPI_F32_SPLITK, out=acc, accumulate=True)
    close(tag + " f32_splitk+acc", acc, 2 * ref, tol)


def test_gemm_bf16_bert_shapes():
    """the real layer shapes at a ragged token count (M = 3*128+40)"""
    M, H, F = 424, 768, 3072
    x = rnd(M, H, dtype=torch.bfloat16, seed=11)
    for (N, K, seed) in [(3 * H, H, 12), (H, H, 13), (F, H, 14), (H, F, 15)]:
        A = x if K == H else rnd(M, K, dtype=torch.bfloat16, seed=seed + 50)
        W = rnd(N, K, dtype=torch.bfloat16, s=0.05, seed=seed)
        close("gemm bert fwd N=%d K=%d" % (N, K), hb.gemm(A, W, M, N, K), A.float() @ W.float().t(), 1e-2)
        dY = rnd(M, N, dtype=torch.bfloat16, seed=seed + 20)
        close("gemm bert dgrad N=%d K=%d" % (N, K), hb.gemm(dY, W, M, K, N, 0, 1), dY.float() @ W.float(), 1e-2)
        close("gemm bert wgrad N=%d K=%d" % (N, K), hb.gemm(dY, A, N, K, M, 1, 1, hb.EPI_F32_SPLITK),
              dY.float().t() @ A.float(), 1e-2)


@pytest.mark.parametrize("H,K,acc", [(768, 4096 + 40, False), (768, 32768, True), (1024, 2048, False)])
def test_wgrad_pair_equals_two_weight_gradients(H, K, acc):
    """nbest_wgrad_pair: the QKV ([3H, H]) and attention-output ([H, H]) weight gradients of a layer in ONE launch (36 / 64 tiles of
    256 x 256, one set of K-splits, one reduce).  Against the fp32 products of the same bf16 operands, and against the two separate
    nbest_gemm launches it replaces (the same numbers up to the fp32 summation order over K-splits); operands with their own leading
    dimensions (dQ|dK|dV rows of 3H, the others H), a token count that is not a multiple of the K-step, and accumulate."""
    bf = torch.bfloat16
    dqkv, x = rnd(K, 3 * H, dtype=bf, seed=1), rnd(K, H, dtype=bf, seed=2)
    dy, ctx = rnd(K, H, dtype=bf, seed=3), rnd(K, H, dtype=bf, seed=4)
    base1, base2 = rnd(3 * H, H, seed=5), rnd(H, H, seed=6)
    o1, o2 = (base1.clone(), base2.clone()) if acc else (None, None)
    g1, g2 = hb.wgrad_pair(dqkv, x, dy, ctx, out1=o1, out2=o2, accumulate=acc)
    r1, r2 = dqkv.float().t() @ x.float(), dy.float().t() @ ctx.float()
    if acc:
        r1, r2 = r1 + base1, r2 + base2
    close("wgrad pair QKV  H=%d K=%d" % (H, K), g1, r1, 2e-5)
    close("wgrad pair attn H=%d K=%d" % (H, K), g2, r2, 2e-5)
    s1 = hb.gemm(dqkv, x, 3 * H, H, K, 1, 1, hb.EPI_F32_SPLITK, out=base1.clone() if acc else None, accumulate=acc)
    s2 = hb.gemm(dy, ctx, H, H, K, 1, 1, hb.EPI_F32_SPLITK, out=base2.clone() if acc else None, accumulate=acc)
    close("wgrad pair vs separate QKV", g1, s1, 2e-6)
    close("wgrad pair vs separate attn", g2, s2, 2e-6)
    # a pair that does not fit (N not a multiple of 256) is refused, not silently mis-tiled
    with pytest.raises(RuntimeError):
        hb.wgrad_pair(rnd(512, 384, dtype=bf), rnd(512, 128, dtype=bf), rnd(512, 128, dtype=bf), rnd(512, 128, dtype=bf))


def _keep_mask(M, N, p, seed, stream):
    """Independent restatement of the counter-based dropout decision (csrc/common.h nb_mix_key / nb_hash32 / nb_keep):
    element idx = m * N + n is kept iff the 16-bit half of hash32((idx >> 1) * 0x9E3779B9 + key) selected by idx & 1 is >= thr16."""
    m64 = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (stream + 1)) & m64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
    key = (z ^ (z >> 31)) & 0xFFFFFFFF
    thr = min(int(round(p * 65536)), 65535)
    idx = torch.arange(M * N, device=DEV, dtype=torch.int64)
    x = ((idx >> 1) * 0x9E3779B9 + key) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    u = torch.where((idx & 1) == 1, x >> 16, x & 0xFFFF)
    return (u >= thr).view(M, N), 65536.0 / (65536 - thr)


@pytest.mark.parametrize("N,K", [(2304, 768), (3072, 768), (768, 3072), (768, 768)])
def test_gemm_bf16_register_epilogues_full_size(N, K):
    """The layer GEMMs at the token count of BASELINE configs[1] minus a ragged tail (the 256-row tile kernels with the
    register epilogue only run from ~1 000 tiles up): every epilogue against fp32 torch on the same bf16 inputs."""
    M = 32768 - 88
    A = rnd(M, K, dtype=torch.bfloat16, s=0.5, seed=31)
    W = rnd(N, K, dtype=torch.bfloat16, s=0.05, seed=32)
    bias = rnd(N, seed=33)
    R = rnd(M, N, dtype=torch.bfloat16, seed=34)
    ref = A.float() @ W.float().t()
    tag = "gemm full-size N=%d K=%d" % (N, K)
    guard = torch.full((M + 512, N), 7.0, dtype=torch.bfloat16, device=DEV)     # rows past M must stay untouched
    out = guard[:M]
    hb.gemm(A, W, M, N, K, out=out)
    close(tag + " none", out, ref, 1e-2)
    assert torch.all(guard[M:] == 7.0), "rows past M were written"
    close(tag + " bias", hb.gemm(A, W, M, N, K, epilogue=hb.EPI_BIAS, bias=bias), ref + bias, 1e-2)
    o, U = hb.gemm(A, W, M, N, K, epilogue=hb.EPI_BIAS_GELU, bias=bias)
    close(tag + " bias_gelu.C", o, gelu(ref + bias), 1e-2)
    gerr = (hb.gelu_d_decode(U) - dgelu(ref + bias)).abs().max().item()
    _log("%-58s abs_err=%.3e" % (tag + " bias_gelu.U 8-bit", gerr))
    assert gerr <= 2.6e-3 + 4e-3, gerr
    close(tag + " res", hb.gemm(A, W, M, N, K, epilogue=hb.EPI_RES, R=R), ref + R.float(), 1e-2)
    y = hb.gemm(A, W, M, N, K, epilogue=hb.EPI_BIAS_DROP_RES, bias=bias, R=R, drop_p=0.1, seed=11, drop_stream=5).float()
    keep, scale = _keep_mask(M, N, 0.1, 11, 5)
    close(tag + " bias_drop_res", y, torch.where(keep, (ref + bias) * scale, torch.zeros_like(ref)) + R.float(), 1e-2)
    Uin = hb.gelu_d_encode(dgelu(rnd(M, N, seed=35)))
    Ud = hb.gelu_d_decode(Uin)
    cs = torch.zeros(N, device=DEV)
    close(tag + " dgelu", hb.gemm(A, W, M, N, K, epilogue=hb.EPI_DGELU, U=Uin, colsum_out=cs), ref * Ud, 1e-2)
    close(tag + " dgelu fused column sums", cs, (ref * Ud).sum(0), 5e-2)


def test_gemm_dropout_mask_is_shared_with_layernorm_bwd():
    M, N, K, p = 256, 256, 64, 0.25
    A = rnd(M, K, dtype=torch.bfloat16, seed=21)
    B = rnd(N, K, dtype=torch.bfloat16, seed=22)
    bias = torch.zeros(N, device=DEV)
    R = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    y0 = hb.gemm(A, B, M, N, K, epilogue=hb.EPI_BIAS_DROP_RES, bias=bias, R=R).float()
    y = hb.gemm(A, B, M, N, K, epilogue=hb.EPI_BIAS_DROP_RES, bias=bias, R=R, drop_p=p, seed=7, drop_stream=3).float()
    dropped = (y == 0) & (y0 != 0)
    frac = dropped.float().mean().item()
    assert abs(frac - p) < 0.02, frac
    kept = ~dropped & (y0 != 0)
    p_eff = round(p * 65536) / 65536
    close("dropout survivors scaled", y[kept], y0[kept] / (1 - p_eff), 2e-2)
    # LN backward regenerates the same mask for the dense-branch gradient
    x = rnd(M, N, dtype=torch.bfloat16, seed=23)
    g = torch.ones(N, device=DEV)
    _, stats = hb.layernorm_fwd(x, g, torch.zeros(N, device=DEV), 1e-12)
    dy = rnd(M, N, dtype=torch.bfloat16, seed=24)
    dx, dxd, *_ = hb.layernorm_bwd(dy, x, stats, g, drop_p=p, seed=7, drop_stream=3)
    assert torch.equal((dxd.float() == 0) & (dx.float() != 0), dropped & (dx.float() != 0))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H", [(37, 768), (300, 1024), (64, 128)])
def test_layernorm_fwd_bwd(dtype, M, H):
    x = rnd(M, H, dtype=dtype, seed=31)
    g, b = 1 + 0.1 * rnd(H, seed=32), 0.1 * rnd(H, seed=33)
    y, stats = hb.layernorm_fwd(x, g, b, 1e-12)
    xr = x.float().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (H,), g, b, 1e-12)
    tol = tol_of(dtype)
    close("ln_fwd %s %dx%d" % (dtype, M, H), y, yr, tol)
    dy = rnd(M, H, dtype=dtype, seed=34)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr2 = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-12)
    yr2.backward(dy.float())
    dx, _, dg, db, dbias = hb.layernorm_bwd(dy, x, stats, g)
    close("ln_bwd dx", dx, xr.grad, tol)
    close("ln_bwd dgamma", dg, gr.grad, 5 * tol)
    close("ln_bwd dbeta", db, br.grad, 5 * tol)
    close("ln_bwd dbias", dbias, dx.float().sum(0), 5 * tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(dtype):
    X = rnd(1000, 2304, dtype=dtype, seed=41)
    close("colsum", hb.colsum(X), X.float().sum(0), 2e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_fwd_bwd(dtype):
    B, S, H, V = 3, 40, 768, 500
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, V, (B, S), generator=g).to(DEV)
    ids[1, 30:] = 0
    ids[2, 20:] = 0                                          # padding rows (row 0 = padding_idx: no gradient)
    seg = (torch.arange(S)[None, :] > 12).long().expand(B, S).contiguous().to(DEV)
    pos = torch.arange(S)[None, :].expand(B, S).contiguous().to(DEV)
    word, tt, pt = rnd(V, H, dtype=dtype, s=0.5, seed=51), rnd(2, H, dtype=dtype, s=0.5, seed=52), rnd(64, H, dtype=dtype, s=0.5, seed=53)
    gam, bet = 1 + 0.1 * rnd(H, seed=54), 0.1 * rnd(H, seed=55)
    out, stats = hb.embed_ln_fwd(ids, seg, pos, word, tt, pt, gam, bet, 1e-12)
    wr, tr, pr = (t.float().clone().requires_grad_(True) for t in (word, tt, pt))
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    e = torch.nn.functional.embedding(ids, wr, padding_idx=0) + tr[seg] + pr[pos]
    ref = torch.nn.functional.layer_norm(e, (H,), gr, br, 1e-12).reshape(B * S, H)
    tol = tol_of(dtype)
    close("embed_fwd", out, ref, tol)
    dout = rnd(B * S, H, dtype=dtype, seed=56)
    ref.backward(dout.float())
    dword, dtt, dpt, dg, db = hb.embed_ln_bwd(ids, seg, pos, word, tt, pt, gam, stats, dout, B, S, word_pad_id=0)
    close("embed_bwd dword", dword, wr.grad, 5 * tol)
    close("embed_bwd dtype", dtt, tr.grad, 5 * tol)
    close("embed_bwd dpos", dpt, pr.grad, 5 * tol)
    close("embed_bwd dgamma", dg, gr.grad, 5 * tol)
    close("embed_bwd dbeta", db, br.grad, 5 * tol)
    assert dword[0].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype,H,roberta", [(torch.float32, 768, False), (torch.bfloat16, 768, True), (torch.bfloat16, 1024, False)])
def test_embed_bwd_segmented_reduce(dtype, H, roberta):
    """The word-table gradient is a segmented reduce over the tokens sorted by id (the index_add of the installed
    BertEmbeddings backward behind /root/reference/models/model.py:43-45, without atomics).  Exercised here: ids that occur once,
    a few times, in EVERY chunk of the sorted order ([SEP]-like: runs spanning dozens of chunks, so the partial-row fix-up sums
    them), runs that end exactly at chunk boundaries, the padding row, the RoBERTa position rule (padding rows keyed to the padding
    position: no gradient), accumulation into existing tables, a host-built permutation against the device-side one - and that two
    runs give the SAME BITS in every output."""
    B, S, V = 37, 53, 400
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(4, V, (B, S), generator=g)
    ids[:, 0] = 2                                            # [CLS]-like: B tokens of one id
    ids[:, 7::9] = 3                                         # [SEP]-like: ~6 per row -> one run of ~220 tokens = ~28 chunks of 8
    for b in range(B):
        ids[b, S - (b % 11):] = 1                            # ragged right padding (id 1 = padding_idx: no gradient)
    ids[5, 10:18] = 77                                       # 8 equal tokens in a row: a run that can sit exactly on a chunk
    ids = ids.to(DEV)
    seg = (torch.arange(S)[None, :] > 12).long().expand(B, S).contiguous().to(DEV)
    if roberta:                                              # model.position_ids_for: cumsum(ids != pad) * (ids != pad) + pad
        nonpad = ids.ne(1).long()
        pos = (torch.cumsum(nonpad, dim=1) * nonpad + 1).contiguous()
        seg = torch.zeros_like(seg)
        n_types, pos_pad = 1, 1
    else:
        pos = torch.arange(S)[None, :].expand(B, S).contiguous().to(DEV)
        n_types, pos_pad = 2, -1
    word, tt, pt = rnd(V, H, dtype=dtype, s=0.5, seed=61), rnd(n_types, H, dtype=dtype, s=0.5, seed=62), rnd(S + 2, H, dtype=dtype, s=0.5, seed=63)
    gam, bet = 1 + 0.1 * rnd(H, seed=64), 0.1 * rnd(H, seed=65)
    out, stats = hb.embed_ln_fwd(ids, seg, pos, word, tt, pt, gam, bet, 1e-5)
    wr, tr, pr = (t.float().clone().requires_grad_(True) for t in (word, tt, pt))
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    e = torch.nn.functional.embedding(ids, wr, padding_idx=1) + tr[seg] + torch.nn.functional.embedding(pos, pr, padding_idx=1 if roberta else None)
    ref = torch.nn.functional.layer_norm(e, (H,), gr, br, 1e-5).reshape(B * S, H)
    dout = rnd(B * S, H, dtype=dtype, seed=66)
    ref.backward(dout.float())
    tol = 5 * tol_of(dtype)
    host_perm = torch.from_numpy(np.argsort(ids.cpu().numpy().ravel(), kind="stable").astype(np.int32)).to(DEV)
    assert torch.equal(host_perm, hb.word_perm(ids)), "host (numpy stable argsort) and device (torch stable sort) permutations differ"
    run = lambda perm: hb.embed_ln_bwd(ids, seg, pos, word, tt, pt, gam, stats, dout, B, S, word_pad_id=1, pos_pad_id=pos_pad, perm=perm)
    a, b_ = run(host_perm), run(None)
    names = ("dword", "dtype", "dpos", "dgamma", "dbeta")
    for n, x, y, want in zip(names, a, b_, (wr.grad, tr.grad, pr.grad, gr.grad, br.grad)):
        assert torch.equal(x, y), "embed_bwd %s is not bit-reproducible" % n
        close("embed_bwd(seg) %s" % n, x, want, tol)
    assert a[0][1].abs().max().item() == 0.0                 # padding row
    # accumulation (the ASR pass on top of the transcript pass): twice the gradient, and still the same bits on a second run
    acc1 = hb.embed_ln_bwd(ids, seg, pos, word, tt, pt, gam, stats, dout, B, S, word_pad_id=1, pos_pad_id=pos_pad, perm=host_perm,
                           accumulate_into=tuple(t.clone() for t in a))
    acc2 = hb.embed_ln_bwd(ids, seg, pos, word, tt, pt, gam, stats, dout, B, S, word_pad_id=1, pos_pad_id=pos_pad, perm=host_perm,
                           accumulate_into=tuple(t.clone() for t in a))
    for n, x, y, one in zip(names, acc1, acc2, a):
        assert torch.equal(x, y), "accumulating embed_bwd %s is not bit-reproducible" % n
        assert torch.equal(x, one + one), "accumulate: %s != 2 x the single pass" % n
    # dropout on: the mask is regenerated from the counter, identically in the two token kernels
    d1 = hb.embed_ln_bwd(ids, seg, pos, word, tt, pt, gam, stats, dout, B, S, word_pad_id=1, pos_pad_id=pos_pad, perm=host_perm, drop_p=0.1, seed=3, drop_stream=5)
    d2 = hb.embed_ln_bwd(ids, seg, pos, word, tt, pt, gam, stats, dout, B, S, word_pad_id=1, pos_pad_id=pos_pad, perm=host_perm, drop_p=0.1, seed=3, drop_stream=5)
    for n, x, y in zip(names, d1, d2):
        assert torch.equal(x, y), n


# ------------------------------------------------------------------------------------------------
def _attn_ref(qkv, mask, B, S, heads):
    H = heads * 64
    q, k, v = qkv.float().reshape(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) / 8.0
    sc = sc.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(sc, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H), torch.logsumexp(sc, -1)


@pytest.mark.parametrize("dtype,S", [(torch.float32, 48), (torch.float32, 37), (torch.bfloat16, 128), (torch.bfloat16, 48),
                                     (torch.bfloat16, 37), (torch.bfloat16, 96), (torch.bfloat16, 100), (torch.bfloat16, 114), (torch.bfloat16, 160), (torch.bfloat16, 256),
                                     (torch.bfloat16, 201), (torch.bfloat16, 230),
                                     # long-sequence kernels (256 < S <= 512): ragged last key block, 9 / 10 / 12 / 16 key blocks
                                     (torch.bfloat16, 257), (torch.bfloat16, 300), (torch.bfloat16, 384), (torch.bfloat16, 512),
                                     (torch.float32, 300)])
def test_attention_fwd_bwd(dtype, S):
    B, heads = 3, 4
    H = heads * 64
    qkv = rnd(B * S, 3 * H, dtype=dtype, s=1.0, seed=61)
    mask = torch.ones(B, S, dtype=torch.uint8, device=DEV)
    mask[1, S - 9:] = 0
    mask[2, S // 2:] = 0
    ctx, lse = hb.attention_fwd(qkv, mask, B, S, heads)
    qr = qkv.float().clone().requires_grad_(True)
    ref, lse_ref = _attn_ref(qr, mask, B, S, heads)
    tol = tol_of(dtype)
    close("attn_fwd ctx %s S=%d" % (dtype, S), ctx, ref, tol)
    close("attn_fwd lse", lse, lse_ref, tol)
    dctx = rnd(B * S, H, dtype=dtype, seed=62)
    ref.backward(dctx.float())
    dbias = torch.zeros(3 * H, device=DEV)
    dqkv = hb.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, heads, dbias=dbias)
    close("attn_bwd fused bias grad %s S=%d" % (dtype, S), dbias, qr.grad.sum(0), 5 * tol)
    d, r = dqkv.float().reshape(B * S, 3, H), qr.grad.reshape(B * S, 3, H)
    for i, nm in enumerate("QKV"):
        close("attn_bwd d%s %s S=%d" % (nm, dtype, S), d[:, i], r[:, i], 2 * tol)


@pytest.mark.parametrize("S", [160, 256])
def test_attention_fwd_bf16_long(S):
    B, heads = 2, 12
    qkv = rnd(B * S, 3 * heads * 64, dtype=torch.bfloat16, seed=63)
    mask = torch.ones(B, S, dtype=torch.uint8, device=DEV)
    mask[1, S - 30:] = 0
    ctx, lse = hb.attention_fwd(qkv, mask, B, S, heads)
    ref, lse_ref = _attn_ref(qkv, mask, B, S, heads)
    close("attn_fwd bf16 S=%d" % S, ctx, ref, 1e-2)
    close("attn_fwd bf16 lse S=%d" % S, lse, lse_ref, 1e-2)


@pytest.mark.parametrize("S", [64, 37, 128, 109, 160, 256, 239, 300, 333])
def test_attention_dropout_fwd_bwd_consistent(S):
    """bf16 MFMA and fp32 VALU kernels must draw the SAME mask from (seed, stream); the backward of
    each must be the gradient of its own forward (checked through the fp32 kernel by finite differences
    on a linear functional).  S = 64: one hash per key pair in the bf16 forward; S = 37 (odd): the per-element
    fallback; S = 160: the five-block backward; S = 300 / 333: the long-sequence kernels (pair hash / per-element)."""
    B, heads, p = 2, 2, 0.2
    H = heads * 64
    qkv32 = rnd(B * S, 3 * H, seed=71)
    mask = torch.ones(B, S, dtype=torch.uint8, device=DEV)
    c32, l32 = hb.attention_fwd(qkv32, mask, B, S, heads, drop_p=p, seed=5, drop_stream=9)
    c16, l16 = hb.attention_fwd(qkv32.bfloat16(), mask, B, S, heads, drop_p=p, seed=5, drop_stream=9)
    close("attn dropout bf16 vs f32 fwd", c16, c32, 3e-2)
    w = rnd(B * S, H, seed=72)
    dq = hb.attention_bwd(qkv32, mask, c32, w, l32, B, S, heads, drop_p=p, seed=5, drop_stream=9)
    dirn = rnd(B * S, 3 * H, seed=73)
    eps = 1e-2
    cp, _ = hb.attention_fwd(qkv32 + eps * dirn, mask, B, S, heads, drop_p=p, seed=5, drop_stream=9)
    cm, _ = hb.attention_fwd(qkv32 - eps * dirn, mask, B, S, heads, drop_p=p, seed=5, drop_stream=9)
    fd = ((cp - cm) * w).sum().item() / (2 * eps)
    an = (dq * dirn).sum().item()
    _log("attn dropout finite-difference %.6f vs analytic %.6f" % (fd, an))
    assert abs(fd - an) <= 2e-2 * max(1.0, abs(fd))
    dq16 = hb.attention_bwd(qkv32.bfloat16(), mask, c16, w.bfloat16(), l16, B, S, heads, drop_p=p, seed=5, drop_stream=9)
    close("attn dropout bf16 vs f32 bwd", dq16, dq, 4e-2)
    # keep words: the forward hands its dropout decisions to the backward (S <= 256): the words must say what the hash says, and
    # the backward that reads them must give the re-hashing backward's result
    c16k, l16k, keep = hb.attention_fwd(qkv32.bfloat16(), mask, B, S, heads, drop_p=p, seed=5, drop_stream=9, want_keep=True)
    assert torch.equal(c16k, c16) and torch.equal(l16k, l16)
    if S <= 256:
        assert keep is not None
        nkb = (S + 31) // 32
        kmask, _ = _keep_mask(B * heads * S, S, p, 5, 9)                 # [(bh, q)][key] decisions of the counter stream
        words = keep.view(B * heads, nkb, nkb * 32)[:, :, :S].to(torch.int64) & 0xFFFFFFFF          # [bh][kb][q]
        bits = (words.unsqueeze(-1) >> torch.arange(32, device=DEV)) & 1                            # [bh][kb][q][j]
        got = bits.permute(0, 2, 1, 3).reshape(B * heads, S, nkb * 32)[:, :, :S].bool()
        bad = (got != kmask.view(B * heads, S, S))
        assert not bad.any(), "keep words disagree with the counter-based decisions: %d of %d, first at %s" % (
            int(bad.sum()), bad.numel(), bad.nonzero()[0].tolist())
        dq16k = hb.attention_bwd(qkv32.bfloat16(), mask, c16, w.bfloat16(), l16, B, S, heads, drop_p=p, seed=5, drop_stream=9, keep=keep)
        # same decisions; the two kernels differ in how the compiler contracts dp * scale - delta into an fma, so single results
        # may differ in the last bf16 bit
        dd = (dq16k.float() - dq16.float()).abs()
        assert dd.max().item() <= 2.0 ** -7 * dq16.float().abs().max().item() and (dd > 0).float().mean().item() < 0.05, (
            "keep-word backward differs from the hashing backward: max %.3e, %.2f %% of the elements" % (dd.max().item(), 100 * (dd > 0).float().mean().item()))
    else:
        assert keep is None


# ------------------------------------------------------------------------------------------------
def test_stc_heads_loss_and_grads_match_oracle(labels):
    from oracle import stc
    B, H = 6, 768
    g = torch.Generator().manual_seed(3)
    hidden = torch.randn(B, 5, H, generator=g)                 # CLS row = [:,0,:]
    heads = stc.OracleHeads(labels.top2bottom, H, labels.n_bottom, 0.0)
    for p_ in heads.parameters():
        torch.nn.init.normal_(p_, std=0.05, generator=g)
    y = torch.zeros(B, labels.n_bottom)
    for b in range(B):
        for t in torch.randperm(labels.n_top, generator=g)[:2].tolist():
            bs = labels.top2bottom[t]
            y[b, bs[int(torch.randint(0, len(bs), (1,), generator=g))]] = 1
    cls = hidden[:, 0, :].clone().requires_grad_(True)
    top, bottoms, final = heads(cls)
    b2t = stc.bottom2top_matrix(labels.top2bottom)
    rec, total, parts = stc.total_loss(top, bottoms, final, y, labels.top2bottom, b2t)
    total.backward()
    dls = hb.DeviceLabelSpace(labels, DEV)
    Wh = torch.cat([heads.top_linear_layer.weight] + [heads.linear_layers["lin_%d" % t].weight for t in labels.multi]).detach().to(DEV)
    bh = torch.cat([heads.top_linear_layer.bias] + [heads.linear_layers["lin_%d" % t].bias for t in labels.multi]).detach().to(DEV)
    hd = hidden.to(DEV).reshape(B * 5, H).contiguous()
    gtop, gbott, gfin, gloss, dcls, dWh, dbh = hb.stc_heads(hd, 5 * H, Wh.contiguous(), bh.contiguous(), dls, y.to(DEV), B, H)
    close("heads top", gtop.cpu(), top, 1e-5)
    close("heads bott", gbott.cpu(), torch.cat([bottoms["lin_%d" % t] for t in labels.multi], 1), 1e-5)
    close("heads final", gfin.cpu(), final, 1e-5)
    close("heads loss parts", gloss[:3].cpu(), torch.stack([parts["bottom_bce"], parts["top_bce"], parts["ce"]]).detach(), 1e-5)
    close("heads dcls", dcls.cpu(), cls.grad, 1e-4)
    dW_ref = torch.cat([heads.top_linear_layer.weight.grad] + [heads.linear_layers["lin_%d" % t].weight.grad for t in labels.multi])
    db_ref = torch.cat([heads.top_linear_layer.bias.grad] + [heads.linear_layers["lin_%d" % t].bias.grad for t in labels.multi])
    close("heads dW", dWh.cpu(), dW_ref, 1e-4)
    close("heads db", dbh.cpu(), db_ref, 1e-4)
    pred = hb.stc_decode(gtop, gbott, dls).cpu()
    ref_dec = stc.decode_indices(top.detach(), {k: v.detach() for k, v in bottoms.items()}, labels.top2bottom, labels.idx2label)
    assert torch.equal(pred.long(), ref_dec)                   # bit-exact label indices


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_stc_heads_dropout_masks(labels, seed):
    """Feature dropout of the STC heads (p = --dropout 0.3): every one of the 11 linears sees its OWN mask of the CLS
    features (/root/reference/models/modules/hierarchical_classifier.py:41,46).  With one utterance whose CLS row is all
    ones, dWh[r][h] / dbh[r] IS the (scaled) mask the backward applied to feature h for the linear that owns row r.
    Checked: kept elements carry 1/(1-p); drop rate ~ p per linear and overall; all rows of a linear share its mask;
    masks of different linears are independent (agreement ~ p^2 + (1-p)^2, never identical); the FORWARD used the same
    masks (scores recomputed from the recovered masks match the kernel's outputs); dcls uses them too."""
    p, H = 0.3, 768
    g = torch.Generator().manual_seed(40 + seed)
    dls = hb.DeviceLabelSpace(labels, DEV)
    R, nt = dls.n_rows, labels.n_top
    Wh = (torch.randn(R, H, generator=g) * 0.05).to(DEV)
    bh = (torch.randn(R, generator=g) * 0.05).to(DEV)
    y = torch.zeros(1, labels.n_bottom)
    y[0, labels.top2bottom[2][3]] = 1
    y[0, labels.top2bottom[4][0]] = 1
    hidden = torch.ones(1, H, device=DEV)
    top, bott, fin, loss, dcls, dWh, dbh = hb.stc_heads(hidden, H, Wh, bh, dls, y.to(DEV), 1, H, drop_p=p, seed=1234 + seed, drop_stream=900)
    top2, bott2, _, _, dcls2, _, _ = hb.stc_heads(hidden, H, Wh, bh, dls, y.to(DEV), 1, H, drop_p=p, seed=1234 + seed, drop_stream=900)
    assert torch.equal(top, top2) and torch.equal(bott, bott2) and torch.equal(dcls, dcls2)          # counter-based: reproducible
    top3, _, _, _, _, _, _ = hb.stc_heads(hidden, H, Wh, bh, dls, y.to(DEV), 1, H, drop_p=p, seed=99 + seed, drop_stream=900)
    assert not torch.equal(top, top3)
    assert dbh.abs().min().item() > 1e-7
    scaled = (dWh / dbh[:, None]).cpu()                         # [R, H]: 0 or 1/(1-p)
    keep_val = 1.0 / (1.0 - p)
    is_keep = (scaled - keep_val).abs() < 1e-3
    is_drop = scaled.abs() < 1e-3
    assert bool((is_keep | is_drop).all()), "mask values other than 0 and 1/(1-p)"
    # rows -> linears: rows [0, nt) are the top linear, then one block per multi-value top label
    blocks, row = [(0, nt)], nt
    for t in labels.multi:
        n = len(labels.top2bottom[t])
        blocks.append((row, row + n))
        row += n
    assert row == R and len(blocks) == 11
    masks = []
    for lo, hi in blocks:
        m = is_keep[lo]
        assert bool((is_keep[lo:hi] == m[None, :]).all()), "rows of one linear must share one mask"
        masks.append(m)
        rate = 1.0 - m.float().mean().item()
        assert abs(rate - p) < 0.07, rate                       # 768 draws: sigma 0.017
    M = torch.stack(masks).float()                              # [11, H]
    assert abs(1.0 - M.mean().item() - p) < 0.02                # 8448 draws: sigma 0.005
    for i in range(11):
        for j in range(i + 1, 11):
            agree = (M[i] == M[j]).float().mean().item()
            assert 0.45 < agree < 0.71, (i, j, agree)            # independent masks agree on p^2 + (1-p)^2 = 0.58
    # forward used the same masks
    Wc, bc = Wh.cpu(), bh.cpu()
    logit = lambda k: (Wc[blocks[k][0]:blocks[k][1]] * (M[k] * keep_val)[None, :]).sum(1) + bc[blocks[k][0]:blocks[k][1]]
    close("heads dropout: top from recovered mask", top.cpu()[0], torch.sigmoid(logit(0)), 1e-5)
    soft = torch.cat([torch.softmax(logit(k), 0) for k in range(1, 11)])
    close("heads dropout: bottoms from recovered masks", bott.cpu()[0], soft, 1e-5)
    # and so did the gradient wrt the CLS row: dcls[h] = sum_r dz[r] W[r][h] mask_{linear(r)}(h) / (1-p)
    want = torch.zeros(H)
    for k, (lo, hi) in enumerate(blocks):
        want += (dbh.cpu()[lo:hi, None] * Wc[lo:hi]).sum(0) * M[k] * keep_val
    close("heads dropout: dcls from recovered masks", dcls.cpu()[0], want, 1e-4)


def test_cls_mse():
    B, S, H = 4, 6, 768
    a, t = rnd(B * S, H, seed=81), rnd(B * 3, H, seed=82)
    da = torch.zeros(B, H, device=DEV)
    dt = torch.zeros(B, H, device=DEV)
    loss = hb.cls_mse(a, S * H, t, 3 * H, B, H, da, dt)
    ar = a.reshape(B, S, H)[:, 0].clone().requires_grad_(True)
    tr = t.reshape(B, 3, H)[:, 0].clone().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(ar, tr)
    ref.backward()
    close("mse loss", loss, ref.detach().reshape(1), 1e-5)
    close("mse da", da, ar.grad, 1e-5)
    close("mse dt", dt, tr.grad, 1e-5)


# ------------------------------------------------------------------------------------------------
def _e4m3(x):
    return x.to(torch.float8_e4m3fn)


@pytest.mark.parametrize("M,N,K", [(200, 256, 192), (424, 768, 768), (300, 3072, 1024), (256, 768, 3072)])
def test_gemm_fp8_forward_epilogues(M, N, K):
    """fp8 forward GEMM (block-scaled MFMA, e4m3 x e4m3, unit block scales): with both operands already e4m3 every product is
    exact in fp32, so the result must match the fp32 matmul of the DEQUANTISED operands to fp32-accumulation accuracy
    (then bf16 output rounding) - this isolates the kernel from the quantisation."""
    g = torch.Generator().manual_seed(M + N + K)
    A8 = _e4m3(torch.randn(M, K, generator=g)).to(DEV)
    W = torch.randn(N, K, generator=g) * 0.03
    s = 2.0 ** math.floor(math.log2(224.0 / W.abs().max().item()))
    W8 = _e4m3(W * s).to(DEV)
    bias = rnd(N, seed=3)
    ref = (A8.float() @ W8.float().t()) / s + bias
    tag = "gemm_fp8[%dx%dx%d]" % (M, N, K)
    close(tag + " bias", hb.gemm_fp8(A8.view(torch.uint8), W8.view(torch.uint8), M, N, K, bias, 1.0 / s), ref, 1e-2)
    out, U, C8 = hb.gemm_fp8(A8.view(torch.uint8), W8.view(torch.uint8), M, N, K, bias, 1.0 / s, epilogue=hb.EPI_BIAS_GELU)
    close(tag + " bias_gelu.C", out, gelu(ref), 1e-2)
    assert (hb.gelu_d_decode(U) - dgelu(ref)).abs().max().item() <= 2.6e-3 + 1e-3
    c8 = C8.view(torch.float8_e4m3fn).float()
    assert torch.equal(c8, _e4m3(out.float().cpu().clamp(-448, 448)).float().to(DEV)) or (c8 - gelu(ref)).abs().max().item() <= 0.07 * gelu(ref).abs().max().item()
    R = rnd(M, N, dtype=torch.bfloat16, seed=4)
    close(tag + " bias_res", hb.gemm_fp8(A8.view(torch.uint8), W8.view(torch.uint8), M, N, K, bias, 1.0 / s, epilogue=hb.EPI_BIAS_DROP_RES, R=R),
          ref + R.float(), 1e-2)
    x = rnd(M, K, dtype=torch.bfloat16, seed=9)
    assert torch.equal(hb.cast_fp8(x).view(torch.float8_e4m3fn).float(), _e4m3(x.float().cpu()).float().to(DEV))     # RNE, like torch


@pytest.mark.parametrize("N,K", [(2304, 768), (768, 768), (3072, 768), (768, 3072)])
def test_gemm_fp8_packed_weights_identical(N, K):
    """nbest_pack_weights_fp8 + nbest_gemm_fp8_args::B_packed: bit-identical to the unpacked call at the training size (256- and 128-column
    tiles take the packed operand) and at a small M (where the other tile width may be chosen and the packed operand ignored)."""
    g = torch.Generator().manual_seed(N + K)
    W8 = _e4m3(torch.randn(N, K, generator=g) * 20).to(DEV).view(torch.uint8)
    Wp, bn = hb.pack_weight_fp8(W8)
    assert Wp is not None and bn in (128, 256) and not torch.equal(Wp, W8)
    bias = rnd(N, seed=4)
    for M in (32768 - 88, 300):
        A8 = _e4m3(torch.randn(M, K, generator=g)).to(DEV).view(torch.uint8)
        R = rnd(M, N, dtype=torch.bfloat16, seed=6)
        for epi, kw in ((hb.EPI_BIAS, {}), (hb.EPI_BIAS_DROP_RES, dict(R=R))):
            ref = hb.gemm_fp8(A8, W8, M, N, K, bias, 0.05, epilogue=epi, **kw)
            got = hb.gemm_fp8(A8, W8, M, N, K, bias, 0.05, epilogue=epi, B_packed=Wp, b_pack_bn=bn, **kw)
            assert torch.equal(got, ref), "packed e4m3 B differs: N=%d K=%d M=%d epi=%d" % (N, K, M, epi)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (768, 768, 1000), (2304, 768, 4096), (768, 3072, 5000)])
def test_wgrad_fp8_transposed_reads(M, N, K):
    """fp8 weight gradient dW = dY8^T . X8 over K token rows (ragged K: zero-filled tail), both operands token-major e4m3 read
    through ds_read_b64_tr_b8: every product is exact in fp32, so the result matches the fp32 matmul of the dequantised operands
    up to the accumulation inside the block-scaled MFMA (measured 1.4e-5 .. 3.1e-5 of the largest entry - the instruction's
    internal sum of 64 products is not a full fp32 chain); split-K + accumulate; gradient scale taken from device float bits"""
    g = torch.Generator().manual_seed(M + N + K)
    dY8 = _e4m3(torch.randn(K, M, generator=g)).to(DEV)
    X8 = _e4m3(torch.randn(K, N, generator=g)).to(DEV)
    ref = dY8.float().t() @ X8.float()
    got = hb.wgrad_fp8(dY8.view(torch.uint8), X8.view(torch.uint8), M, N, K)
    close("wgrad_fp8[%dx%d K=%d]" % (M, N, K), got, ref, 1e-4)
    amax = torch.tensor([3.0], device=DEV).view(torch.int32)             # gradient scale 2^floor(log2(56/3)) = 16 (common.h fp8_gscale_of)
    got2 = hb.wgrad_fp8(dY8.view(torch.uint8), X8.view(torch.uint8), M, N, K, a_amax=amax, out=got.clone(), accumulate=True)
    close("wgrad_fp8 scaled + accumulate", got2, ref * (1 + 1 / 16.0), 1e-4)
    # a zero / non-finite amax (nothing recorded yet, an overflowed pass) means scale 1, not 0 or inf
    for bad in (0.0, float("inf"), float("nan"), 1e-40):
        amax = torch.tensor([bad], device=DEV).view(torch.int32)
        got3 = hb.wgrad_fp8(dY8.view(torch.uint8), X8.view(torch.uint8), M, N, K, a_amax=amax)
        close("wgrad_fp8 amax=%r -> scale 1" % bad, got3, ref, 1e-4)


@pytest.mark.parametrize("H,K", [(768, 5000), (1024, 2048), (768, 32768)])
def test_wgrad_fp8_pair(H, K):
    """nbest_wgrad_fp8_pair: the e4m3 Q|K|V ([3H, H]) and attention-output ([H, H]) weight gradients of a layer in one launch, each with
    its OWN gradient scale, against the fp32 products of the dequantised operands (tolerance of the single-problem test) - rows of dQ|dK|dV
    with leading dimension 3H, ragged K, accumulate into existing gradients."""
    g = torch.Generator().manual_seed(H + K)
    dqkv8 = _e4m3(torch.randn(K, 3 * H, generator=g)).to(DEV)
    x8 = _e4m3(torch.randn(K, H, generator=g)).to(DEV)
    dy8 = _e4m3(torch.randn(K, H, generator=g)).to(DEV)
    ctx8 = _e4m3(torch.randn(K, H, generator=g)).to(DEV)
    r1, r2 = dqkv8.float().t() @ x8.float(), dy8.float().t() @ ctx8.float()
    u8 = lambda t: t.view(torch.uint8)
    g1, g2 = hb.wgrad_fp8_pair(u8(dqkv8), u8(x8), u8(dy8), u8(ctx8))
    close("wgrad_fp8_pair QKV  H=%d K=%d" % (H, K), g1, r1, 1e-4)
    close("wgrad_fp8_pair attn H=%d K=%d" % (H, K), g2, r2, 1e-4)
    a1 = torch.tensor([3.0], device=DEV).view(torch.int32)     # scale 16
    a2 = torch.tensor([0.4], device=DEV).view(torch.int32)     # scale 2^floor(log2(56 / 0.4)) = 128
    h1, h2 = hb.wgrad_fp8_pair(u8(dqkv8), u8(x8), u8(dy8), u8(ctx8), amax_a=a1, amax_b=a2, outs=(g1.clone(), g2.clone()), accumulate=True)
    close("wgrad_fp8_pair scaled + accumulate QKV", h1, r1 * (1 + 1 / 16.0), 1e-4)
    close("wgrad_fp8_pair scaled + accumulate attn", h2, r2 * (1 + 1 / 128.0), 1e-4)


@pytest.mark.parametrize("N,K", [(2304, 768), (3072, 768), (768, 3072), (768, 768), (1024, 1024)])
def test_gemm_packed_weights_identical(N, K):
    """nbest_pack_weights + nbest_gemm_args::B_packed: the B tile of the k-contiguous kernels staged by a linear LDS-DMA copy from a
    pre-packed weight matrix.  Same bytes in the same LDS image, so the result must be BIT-identical to the unpacked call - at the
    training size (256 x 256 / 256 x 192 tiles take the packed operand) and at a small M (another kernel: the packed operand is ignored)."""
    bf = torch.bfloat16
    W = rnd(N, K, dtype=bf, s=0.05, seed=3)
    Wp, bn = hb.pack_weight(W)
    assert Wp is not None and bn in (192, 256)
    assert not torch.equal(Wp.view(-1), W.view(-1))                 # it IS a permutation of the rows / chunks ...
    assert torch.equal(Wp.view(-1).sort().values, W.view(-1).sort().values)    # ... of the same elements
    bias = rnd(N, seed=4)
    for M in (32768 - 88, 424):
        A = rnd(M, K, dtype=bf, seed=5)
        R = rnd(M, N, dtype=bf, seed=6)
        for epi, kw in ((hb.EPI_BIAS, dict(bias=bias)), (hb.EPI_RES, dict(R=R)), (hb.EPI_NONE, {})):
            ref = hb.gemm(A, W, M, N, K, epilogue=epi, **kw)
            got = hb.gemm(A, W, M, N, K, epilogue=epi, B_packed=Wp, b_pack_bn=bn, **kw)
            assert torch.equal(got, ref), "packed B differs: N=%d K=%d M=%d epi=%d" % (N, K, M, epi)
        close("gemm packed N=%d K=%d M=%d" % (N, K, M), got, A.float() @ W.float().t(), 1e-2)


def test_sparse_row_exchange_kernels():
    """nbest_rows_gather / _zero / _add (the word-embedding rows data-parallel ranks exchange, trainer.GradReducer) against the
    torch index ops they replace: bit-exact (copies and one fp32 addition per element), padding slots (id -1, zeros) included."""
    V, H, n, cap = 5000, 768, 37, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    table = torch.randn(V, H, generator=g).to(DEV)
    rows = torch.randperm(V, generator=g)[:n].sort().values.to(DEV)
    ids, vals = hb.rows_gather(table, rows, cap)
    assert torch.equal(ids[:n], rows) and bool((ids[n:] == -1).all())
    assert torch.equal(vals[:n], table.index_select(0, rows)) and bool((vals[n:] == 0).all())
    ref = table.clone()
    ref.index_fill_(0, rows, 0.0)
    hb.rows_zero(table, rows)
    assert torch.equal(table, ref)
    other = torch.randn(cap, H, generator=g).to(DEV)
    ref.index_add_(0, rows, other[:n])
    hb.rows_add(table, ids, other)                      # the padding slots (id -1) must be skipped
    assert torch.equal(table, ref)
    e_ids, e_vals = hb.rows_gather(table, rows[:0], 1)  # a rank whose shard touches no row still sends one padding slot
    assert e_ids.tolist() == [-1] and float(e_vals.abs().max()) == 0.0


def test_transposed_weight_copies():
    """k-contiguous copies of the weight matrices for the dgrad GEMMs: bf16 transpose (nbest_transpose_weights) and e4m3 copy +
    its transpose (nbest_quantize_weights_fp8), on matrices that take the 16-byte tile path (dimensions multiples of 64) and on
    ones that take the element-wise path (ragged) - bit-exact against torch."""
    import ctypes as C
    shapes = [(128, 192), (100, 72), (64, 256), (70, 64)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += (r * c + 63) // 64 * 64
    g = torch.Generator().manual_seed(7)
    master = torch.randn(total, generator=g).to(DEV)
    arr = (hb.MatrixDesc * len(shapes))()
    t = 0
    for i, ((r, c), off) in enumerate(zip(shapes, offs)):
        arr[i].offset, arr[i].rows, arr[i].cols, arr[i].tile_start = off, r, c, t
        t += ((r + 63) // 64) * ((c + 63) // 64)
    descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    w16 = master.bfloat16()
    w16t = torch.zeros_like(w16)
    hb.check(hb.lib().nbest_transpose_weights(hb.ptr(w16), hb.ptr(w16t), hb.ptr(descs), len(shapes), t, hb.stream_ptr()), "transpose_weights")
    w8, w8t = torch.zeros(total, dtype=torch.uint8, device=DEV), torch.zeros(total, dtype=torch.uint8, device=DEV)
    inv = torch.zeros(len(shapes), device=DEV)
    ws = torch.zeros(4 * len(shapes) + 16, dtype=torch.uint8, device=DEV)
    hb.check(hb.lib().nbest_quantize_weights_fp8(hb.ptr(master), hb.ptr(w8), hb.ptr(w8t), hb.ptr(descs), len(shapes), t, hb.ptr(inv),
                                                 hb.ptr(ws), ws.numel(), hb.stream_ptr()), "quantize_weights_fp8")
    for i, ((r, c), off) in enumerate(zip(shapes, offs)):
        m = master[off:off + r * c].view(r, c)
        assert torch.equal(w16t[off:off + r * c].view(c, r), m.bfloat16().t()), shapes[i]
        s = 2.0 ** math.floor(math.log2(224.0 / m.abs().max().item()))
        assert abs(inv[i].item() * s - 1.0) < 1e-6
        q = _e4m3((m * s).cpu()).view(torch.uint8).to(DEV)
        assert torch.equal(w8[off:off + r * c].view(r, c), q), shapes[i]
        assert torch.equal(w8t[off:off + r * c].view(c, r), q.t()), shapes[i]


def test_decode_into_pinned_host_rows_and_stream_stamp(labels):
    """nbest_stc_decode with `pred` in mapped pinned HOST memory + nbest_stream_stamp behind it (the per-step prediction hand-off of
    trainer.MetricsPipe: no hipMemcpy, no event): once the host sees the stamp, the rows it reads are the rows a device-side decode
    of the same scores gives - over many back-to-back steps, alternating two buffers as the epoch loop does."""
    import time
    dls = hb.DeviceLabelSpace(labels, DEV)
    B = 256
    bufs = [torch.full((B, labels.n_top), -7, dtype=torch.int32).pin_memory() for _ in range(2)]
    flags = torch.zeros(2, dtype=torch.int32).pin_memory()
    want = []
    for step in range(1, 41):
        g = torch.Generator(device="cpu").manual_seed(step)
        top = torch.rand(B, labels.n_top, generator=g).to(DEV)
        bott = torch.rand(B, dls.n_rows - labels.n_top, generator=g).to(DEV)      # the multi-value heads' scores, concatenated
        t = step & 1
        hb.stc_decode(top, bott, dls, out=bufs[t])
        hb.stream_stamp(flags[t:t + 1], step)
        ref = hb.stc_decode(top, bott, dls)                 # the same decode into device memory
        want.append((t, step, ref))
        if step >= 2:                                        # consume the PREVIOUS step, as MetricsPipe does
            pt, ps, pref = want[-2]
            t0 = time.time()
            while int(flags[pt]) != ps:
                assert time.time() - t0 < 30.0, "stamp %d never arrived" % ps
                time.sleep(0.0002)
            assert torch.equal(bufs[pt], pref.cpu()), "step %d: host rows differ from the device decode" % ps
    torch.cuda.synchronize()


def test_fp8_amax_fold():
    """nbest_fp8_amax_fold: out[t] = max over the 16 slot words of tensor t (float bits of non-negative values order like the
    floats), slots zeroed; words outside the 16 slot positions are not touched"""
    n, W = 7, hb.AMAX_TENSOR_WORDS
    g = torch.Generator(device="cpu").manual_seed(3)
    slots = torch.zeros(n, W, dtype=torch.float32)
    pos = torch.arange(16) * (W // 16)
    vals = torch.rand(n, 16, generator=g) * torch.tensor([1e-3, 1.0, 37.0, 448.0, 1e4, 0.0, 5.0]).view(n, 1)
    slots[:, pos] = vals
    slots[:, 5] = 123.0                                      # not a slot position: must survive
    d = slots.view(torch.int32).reshape(-1).to(DEV)
    out = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    hb.fp8_amax_fold(d, out)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.float32).cpu(), vals.max(dim=1).values)
    left = d.view(torch.float32).view(n, W).cpu()
    assert (left[:, pos] == 0).all() and (left[:, 5] == 123.0).all()
