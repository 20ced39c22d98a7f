#!/usr/bin/env python3
"""soak: a few hundred full-size training steps (bf16, or the fp8w mode) with dropout on fresh synthetic batches - loss must fall
and every parameter / moment stay finite.  python tools/soak.py [steps] [bf16|fp8w] [seed]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, synth
from nbest_amd.model import NBestSTCModel
from nbest_amd.optim import HipBertAdam
from nbest_amd.trainer import train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fp8 = len(sys.argv) > 2 and sys.argv[2] == "fp8w"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 999
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
cfg = ncfg.bert_base()
m = NBestSTCModel(cfg, labels, device="cuda:0", compute_dtype=torch.bfloat16, dropout=0.3, seed=seed, fp8_forward=fp8)
m.load_reference_state(synth.model_state(cfg, labels, seed=seed))
m.train()
opt = HipBertAdam(m, lr=3e-5, bert_lr=3e-5, warmup=0.1, t_total=steps)
pool = []
for s in range(8):          # a small pool of batches whose labels depend on the tokens, so there is something to learn
    b = synth.nbest_batch(cfg, labels, 256, 128, n_best=5, seed=1000 + s, ragged=True)
    pool.append({k: torch.from_numpy(v).cuda() for k, v in b.items()})
hist = []
for s in range(steps):
    out = train_step(m, opt, pool[s % len(pool)], add_segment_ids=True)
    if s % 20 == 0 or s == steps - 1:
        hist.append((s, out["loss_parts"].sum().item() / 256))
        print("step %4d  loss/utt %.4f" % hist[-1], flush=True)
a = m.arena
ok = all(torch.isfinite(t).all().item() for t in (a.p, a.m, a.v, a.g))
print("finite:", ok, " first %.3f -> last %.3f" % (hist[0][1], hist[-1][1]))
assert ok and hist[-1][1] < 0.5 * hist[0][1]
print("OK")
