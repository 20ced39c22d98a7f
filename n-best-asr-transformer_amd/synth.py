"""Deterministic synthetic weights and n-best batches (numpy only; no torch, no GPU).

Used by bench.py (throughput on synthetic n-best sequences, BASELINE.json configs), by the parity
tests and by tests/golden/make_golden.py, so that ONLY outputs need to be committed as fixtures.
Layout of a batch follows /root/reference/utils/bert_xlnet_inputs.py:77-102:
``[CLS] a.. [SEP] h1 [SEP] h2 .. hn [SEP] <pad>..`` with segment 0 up to and excluding the first
separator, 1 after, 0 on pads; labels are multi-hot [B, n_bottom] with at most one active bottom
label per multi-value top label (/root/reference/utils/STC_util.py:34).
"""
import zlib

import numpy as np


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def encoder_param_shapes(cfg):
    H, F = cfg.hidden_size, cfg.intermediate_size
    shapes = [("embeddings.word_embeddings.weight", (cfg.vocab_size, H)),
              ("embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H)),
              ("embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H)),
              ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))]
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layer.%d." % i
        for nm in ("query", "key", "value"):
            shapes += [(p + "attention.self.%s.weight" % nm, (H, H)), (p + "attention.self.%s.bias" % nm, (H,))]
        shapes += [(p + "attention.output.dense.weight", (H, H)), (p + "attention.output.dense.bias", (H,)),
                   (p + "attention.output.LayerNorm.weight", (H,)), (p + "attention.output.LayerNorm.bias", (H,)),
                   (p + "intermediate.dense.weight", (F, H)), (p + "intermediate.dense.bias", (F,)),
                   (p + "output.dense.weight", (H, F)), (p + "output.dense.bias", (H,)),
                   (p + "output.LayerNorm.weight", (H,)), (p + "output.LayerNorm.bias", (H,))]
    shapes += [("pooler.dense.weight", (H, H)), ("pooler.dense.bias", (H,))]
    return shapes


def head_param_shapes(labels, in_dim):
    shapes = [("top_linear_layer.weight", (labels.n_top, in_dim)), ("top_linear_layer.bias", (labels.n_top,))]
    for t in labels.multi:
        n = len(labels.top2bottom[t])
        shapes += [("linear_layers.lin_%d.weight" % t, (n, in_dim)), ("linear_layers.lin_%d.bias" % t, (n,))]
    return shapes


def _fill(name, shape, seed, std):
    g = _rng(seed, name)
    if name.endswith("LayerNorm.weight"):
        return (1.0 + 0.05 * g.standard_normal(shape)).astype(np.float32)
    if name.endswith("bias"):
        return (std * g.standard_normal(shape)).astype(np.float32)
    return (std * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)


def model_state(cfg, labels, seed=999, std=0.02, head_std=0.05):
    """name -> float32 ndarray with the reference's state_dict keys (bert_encoder.*, clf.*)."""
    sd = {}
    for n, s in encoder_param_shapes(cfg):
        sd["bert_encoder." + n] = _fill(n, s, seed, std)
    for n, s in head_param_shapes(labels, cfg.hidden_size):
        sd["clf." + n] = _fill("clf." + n, s, seed, head_std)
    return sd


def pretrained_like(sd, cfg, seed=999, n_dims=6, ln_gain=10.0, col_gain=20.0):
    """Give a random-init state the statistics that set pretrained BERT / XLM-R apart from std-0.02 noise: a handful of OUTLIER
    feature dimensions - LayerNorm gains x ``ln_gain`` there (every LayerNorm of the stack), the same columns of the word /
    position tables and of the dense-output biases x ``col_gain``.  Those dimensions then carry activations one to two orders
    of magnitude above the rest through the whole residual stream: the distribution the unit-scale e4m3 activations, the 8-bit
    GELU' and the per-tensor gradient scales of the fp8 mode have to survive.  In place; returns the outlier dimensions."""
    H = cfg.hidden_size
    dims = np.sort(_rng(seed, "outlier-dims").choice(H, size=n_dims, replace=False))
    for n, v in sd.items():
        if not n.startswith("bert_encoder."):
            continue
        if n.endswith("LayerNorm.weight"):
            v[dims] *= ln_gain
        elif n.endswith(("word_embeddings.weight", "position_embeddings.weight")):
            v[:, dims] *= col_gain
        elif n.endswith(("attention.output.dense.bias", ".output.dense.bias")) and v.shape[0] == H:
            v[dims] *= col_gain
    return dims


def nbest_batch(cfg, labels, B, S, n_best=5, seed=999, ragged=False, trans_len=None):
    """Synthetic padded hypothesis batch.  Returns dict of int64/float32 ndarrays:
    ids[B,S], seg[B,S], labels[B,n_bottom] (+ tids/tseg [B,trans_len] when trans_len)."""
    g = _rng(seed, "batch/%d/%d/%d" % (B, S, n_best))
    pad, cls, sep = cfg.pad_token_id, cfg.cls_token_id, cfg.sep_token_id
    lo = 1000 if cfg.family == "bert" else 4
    ids = np.full((B, S), pad, np.int64)
    seg = np.zeros((B, S), np.int64)

    def fill(row_ids, row_seg, length, n_hyp):
        # [CLS] a (la tokens) [SEP] then n_hyp hypotheses each followed by [SEP]; total == length
        la_max = max(1, min(27, length - 2 - 2 * n_hyp))
        la = int(g.integers(min(4, la_max), la_max + 1))
        body = length - 2 - la - n_hyp            # tokens shared by the hypotheses
        cuts = np.sort(g.integers(0, body + 1, n_hyp - 1)) if n_hyp > 1 else np.array([], np.int64)
        lens = np.diff(np.concatenate([[0], cuts, [body]]))
        toks = g.integers(lo, cfg.vocab_size, length)
        toks[0] = cls
        p = 1 + la
        toks[p] = sep
        row_seg[p:length] = 1
        p += 1
        for hl in lens:
            p += int(hl)
            toks[p] = sep
            p += 1
        assert p == length
        row_ids[:length] = toks

    for b in range(B):
        length = S if (not ragged or b == 0) else int(g.integers(S // 2, S + 1))
        fill(ids[b], seg[b], length, n_best)
    y = np.zeros((B, labels.n_bottom), np.float32)
    for b in range(B):
        k = int(g.choice([1, 2, 3], p=[0.69, 0.30, 0.01]))
        tops = g.choice(labels.n_top, size=k, replace=False)
        for t in tops:
            bs = labels.top2bottom[int(t)]
            y[b, bs[int(g.integers(0, len(bs)))]] = 1.0
    out = dict(ids=ids, seg=seg, labels=y)
    if trans_len:
        tids = np.full((B, trans_len), pad, np.int64)
        tseg = np.zeros((B, trans_len), np.int64)
        for b in range(B):
            length = trans_len if (not ragged or b == 0) else int(g.integers(trans_len // 2, trans_len + 1))
            fill(tids[b], tseg[b], length, 1)
        out.update(tids=tids, tseg=tseg)
    return out
