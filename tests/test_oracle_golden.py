"""CPU: the oracle (oracle/) reproduces the committed reference outputs (tests/golden/*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case, case_inputs


def _oracle_run(meta, labels):
    from oracle.encoder import EncoderConfig
    from oracle.model import OracleModel
    from oracle import stc
    cfg, sd, batch = case_inputs(meta, labels)
    ocfg = EncoderConfig(**{k: v for k, v in cfg.to_dict().items() if k in EncoderConfig.__dataclass_fields__})
    m = OracleModel(ocfg, labels.top2bottom, labels.n_bottom, 0.0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.train()
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    seg = t["seg"] if meta["seg"] else None
    top, bottoms, final, asr, tr = m(t["ids"], t["tids"], seg_ids=seg, trans_seg_ids=t["tseg"])
    b2t = stc.bottom2top_matrix(labels.top2bottom)
    rec, total, parts = stc.total_loss(top, bottoms, final, t["labels"], labels.top2bottom, b2t, asr, tr, meta["add_l2"])
    total.backward()
    return m, top, bottoms, final, asr, tr, rec, total


@pytest.mark.parametrize("name", ["bert_L2", "bert_L2_noseg", "xlmr_L2", "bert_L12_S256"])
def test_oracle_matches_reference_outputs(name, labels):
    from oracle import stc
    from oracle.bertadam import OracleBertAdam
    meta, z = load_case(name)
    m, top, bottoms, final, asr, tr, rec, total = _oracle_run(meta, labels)
    np.testing.assert_allclose(top.detach().numpy(), z["top"], atol=2e-6)
    np.testing.assert_allclose(final.detach().numpy(), z["final"], atol=2e-6)
    np.testing.assert_allclose(asr.detach().numpy(), z["asr_cls"], atol=2e-5)
    np.testing.assert_allclose(tr.detach().numpy(), z["trans_cls"], atol=2e-5)
    assert abs(total.item() - float(z["loss_total"])) < 2e-4 * abs(float(z["loss_total"]))
    assert abs(rec - float(z["loss_record"])) < 2e-4 * abs(float(z["loss_record"]))
    dec = stc.decode_indices(top.detach(), {k: v.detach() for k, v in bottoms.items()}, labels.top2bottom, labels.idx2label)
    assert np.array_equal(dec.numpy(), z["decode"])          # bit-exact label indices
    named = dict(m.named_parameters())
    for key in z.files:
        if key.startswith("gnorm/"):
            g = named[key[6:]].grad
            assert abs(g.norm().item() - float(z[key])) <= 2e-4 * max(1e-3, float(z[key])), key
        if key.startswith("grad/"):
            g = named[key[5:]].grad
            got = g.reshape(-1, g.shape[-1])[:8, :64] if g.dim() > 1 else g[:64]
            np.testing.assert_allclose(got.numpy(), z[key], atol=5e-5 * max(1.0, float(np.abs(z[key]).max())))
    before = {n: p.detach().clone() for n, p in named.items()}
    opt = OracleBertAdam(list(named.items()), lr=5e-4, bert_lr=3e-5, warmup=0.1, t_total=int(z["t_total"]))
    opt.step(); opt.step()
    for key in z.files:
        if key.startswith("delta/"):
            d = named[key[6:]].detach() - before[key[6:]]
            got = d.reshape(-1, d.shape[-1])[:8, :64] if d.dim() > 1 else d[:64]
            np.testing.assert_allclose(got.numpy(), z[key], atol=3e-7)


def test_bf16_storage_leg_reproduces_committed_floor(labels):
    """the bf16-storage leg of the oracle (oracle/bf16sim.py) gives the committed noise floors of a case again (they bound
    the HIP bf16 path in tests/test_model_gpu.py); rounding decisions can flip with the CPU's summation order, hence 30 %"""
    from oracle import bf16sim, stc
    meta, z = load_case("bert_L2")
    m, top, bottoms, final, asr, tr, rec, total = _oracle_run(meta, labels)
    cfg, sd, batch = case_inputs(meta, labels)
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    stop, sbot, sfin, sasr, str_ = bf16sim.forward(m, t["ids"], t["tids"], seg_ids=t["seg"], trans_seg_ids=t["tseg"])
    for key, a, b in (("top", stop, top), ("final", sfin, final), ("asr_cls", sasr, asr), ("trans_cls", str_, tr)):
        got = (a - b).abs().max().item()
        assert 0.7 * float(z["floor0/" + key][0]) <= got <= 1.3 * float(z["floor0/" + key][0]), (key, got, float(z["floor0/" + key][0]))
        assert got <= 1.05 * float(z["floor/" + key][0])          # the committed floor is the maximum over five draws of this leg
    # and it is a bf16-sized perturbation, not a different function: far above fp32 noise, far below the signal
    assert 1e-4 < (stop - top).abs().max().item() < 2e-2


ALL_CASES = ["bert_L2", "bert_L2_noseg", "xlmr_L2", "bert_L12", "bert_L12_S256", "xlmr_L12", "xlmrL_L4_S256", "bert_L4_outliers",
             "xlmrL_L24_S256", "bert_L4_outliers_big"]


@pytest.mark.parametrize("name", ALL_CASES)
def test_8bit_gelu_derivative_costs_no_gradient_accuracy(name):
    """The bf16 path keeps gelu'(u) for the backward in 8-bit fixed point (csrc/common.h gd_pack4) and its parity bar is the
    noise floor of an oracle leg with the SAME 8-bit rounding (`floor/`).  So that the bar cannot hide a loss against plain bf16
    storage, every case also commits the leg with gelu' kept in bf16 (`floorb/`): per encoder-layer matrix the gradient
    noise-to-signal of the 8-bit leg must not exceed the bf16 leg's by more than the draw-to-draw spread of the two legs
    (median over the matrices within 5 %, worst matrix within 20 %), and the score floors must agree."""
    meta, z = load_case(name)
    # (floor0/ = the unjittered draw of the leg: floor/ itself is the maximum over five draws since round 4)
    dense = [k[len("floor0/ns/"):] for k in z.files if k.startswith("floor0/ns/bert_encoder.encoder.") and k.endswith(".weight")
             and "LayerNorm" not in k]
    assert len(dense) >= 12
    ratios = sorted(float(z["floor0/ns/" + n][0]) / max(float(z["floorb/ns/" + n][0]), 1e-30) for n in dense)
    assert ratios[len(ratios) // 2] <= 1.05 and ratios[-1] <= 1.2, (ratios[len(ratios) // 2], ratios[-1])
    for k in ("top", "final", "bottoms"):
        assert float(z["floor0/" + k][0]) <= 1.25 * float(z["floorb/" + k][0]) + 1e-6, k
        assert float(z["floor/" + k][0]) >= float(z["floor0/" + k][0])


@pytest.mark.parametrize("name", ALL_CASES)
def test_fp8_floor_is_committed_for_every_case(name):
    """every reference-generated case carries the fp8 leg's floors (scores, CLS rows, loss, per-tensor gradient statistics): the
    bar the fp8w path (BASELINE configs[4]) is held to in tests/test_model_gpu.py - and they are fp8-sized: above the bf16 floor,
    far below the signal"""
    meta, z = load_case(name)
    for k in ("top", "final", "bottoms", "asr_cls", "loss_total"):
        assert ("floor8/" + k) in z.files
        if k != "loss_total":        # (the loss is ONE scalar: its error in either leg is a single draw, see tests/test_model_gpu.py _loss_bound)
            # (outlier models: the bf16 leg's storage noise on 100+-sized activations already dominates, so the two floors are
            # draws of comparable noise - 0.7; elsewhere the fp8 floor sits above the bf16 one)
            assert float(z["floor8/" + k][0]) >= (0.7 if meta.get("outliers") else 0.8) * float(z["floor/" + k][0]), k
    assert float(z["floor8/top"][0]) < 0.15 * max(1.0, meta["L"] / 12.0) ** 0.5      # (storage noise grows like sqrt(depth): 24 layers 0.18)
    if name == "bert_L4_outliers_big":
        # the yardstick leg with UNIT-scale e4m3 activations (the fp8 forward before round 4).  Activations beyond 448 saturate in it:
        # several times the error of the per-tensor scales - what the scale is worth.  (On bert_L4_outliers, amax 260, nothing
        # saturates and the two legs are two draws of the same noise: a power-of-two scale leaves every normal e4m3 value unchanged.)
        assert float(z["act_amax"].max()) > 448.0
        assert float(z["floor8u/final"][0]) >= 2.5 * float(z["floor8/final"][0])
        assert float(z["floor8u/asr_cls"][0]) >= 2.5 * float(z["floor8/asr_cls"][0])
    n8 = [k for k in z.files if k.startswith("floor8/ns/")]
    assert len(n8) == len([k for k in z.files if k.startswith("floor/ns/")]) > 20
    assert len(z["floor/loss_draws"]) == len(z["floor8/loss_draws"]) == 5       # the loss bar: five draws of each leg
    assert any(k.startswith("floor8/samp/") for k in z.files) and "floor8/wordgrad" in z.files


def test_reference_known_answers(labels):
    from oracle import stc
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    k = kat["onehot_to_scalar"]
    assert stc.class_index(torch.tensor(k["inp"], dtype=torch.float32)).tolist() == k["out"] == [2, 1, 2, 2, 0]
    for c in kat["update_f1"]:
        base = (0, 0, 0) if c["pred"] else (2, 3, 4)
        assert list(stc.update_f1(c["pred"], c["gold"], *base)) == c["out"]
    for c in kat["compute_f1"]:
        assert list(stc.compute_f1(*c["inp"])) == pytest.approx(c["out"])
    assert stc.bottom2top_matrix(labels.top2bottom).argmax(dim=1).tolist() == kat["bottom2top_argmax"]
    assert labels.bottom2top == kat["bottom2top_argmax"]
    assert labels.n_bottom == 161 and labels.n_top == 30 and labels.n_head_rows == 171
