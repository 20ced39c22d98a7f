#!/usr/bin/env python3
"""timing experiment: how much of the BertAdam step hides behind the backward when it runs on a second stream?
(the whole optimizer is launched after the FIRST backward chunk - numerically meaningless, timing only)"""
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

dev = torch.device("cuda", 0)
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
cfg = ncfg.bert_base()
model = NBestSTCModel(cfg, labels, device=dev, compute_dtype=torch.bfloat16, dropout=0.3, seed=999)
model.load_reference_state(synth.model_state(cfg, labels, seed=999))
model.train()
b = synth.nbest_batch(cfg, labels, 256, 128, n_best=5, seed=999)
batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
optim = HipBertAdam(model, lr=3e-5, bert_lr=3e-5, warmup=0.1, t_total=100000)
side = torch.cuda.Stream()
chunks = [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 12)]


def plain():
    model.forward_backward(batch["ids"], batch["labels"], seg_ids=batch["seg"])
    optim.step()


def overlapped():
    def cb(lo, hi):
        if lo == 10:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                optim.step()
    model.forward_backward(batch["ids"], batch["labels"], seg_ids=batch["seg"], chunks=chunks, on_chunk_done=cb)
    torch.cuda.current_stream().wait_stream(side)


def chunked_only():
    model.forward_backward(batch["ids"], batch["labels"], seg_ids=batch["seg"], chunks=chunks, on_chunk_done=lambda lo, hi: None)
    optim.step()


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print("plain (one backward call, optimizer after)     %.3f ms" % timeit(plain))
print("chunked backward, optimizer after              %.3f ms" % timeit(chunked_only))
print("optimizer on a side stream during the backward %.3f ms" % timeit(overlapped))
