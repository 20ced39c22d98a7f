#!/usr/bin/env python3
"""experiment: capture one whole training step (forward + losses + backward + BertAdam) in a HIP graph and compare the
replay time with eager launches.  python tools/graph_try.py"""
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, synth
from nbest_amd.model import NBestSTCModel
from nbest_amd.optim import HipBertAdam
from nbest_amd.trainer import train_step

dev = torch.device("cuda", 0)
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
cfg = ncfg.bert_base()
model = NBestSTCModel(cfg, labels, device=dev, compute_dtype=torch.bfloat16, dropout=0.3, seed=999)
model.load_reference_state(synth.model_state(cfg, labels, seed=999))
model.train()
b = synth.nbest_batch(cfg, labels, 256, 128, n_best=5, seed=999)
batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
optim = HipBertAdam(model, lr=3e-5, bert_lr=3e-5, warmup=0.1, t_total=100000)


def step():
    return train_step(model, optim, batch, add_segment_ids=True)


def timeit(f, n=20):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for _ in range(5):
    step()
print("eager  %.3f ms/step" % timeit(step), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = step()
    print("graph  %.3f ms/step (replay; NOTE: dropout seeds and the LR schedule are frozen in a replay)" % timeit(g.replay), flush=True)
except Exception as e:
    print("capture failed:", repr(e)[:400])
