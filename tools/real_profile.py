#!/usr/bin/env python3
"""where the HOST time of the real-data loop goes: cProfile of trainer.train_epoch over the tiled valid split (bench.py --real's
workload), main thread only + wall time per step.  python tools/real_profile.py [n_utts]"""
import cProfile
import json
import os
import pstats
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, inputs, synth, trainer
from nbest_amd.model import NBestSTCModel
from nbest_amd.optim import HipBertAdam

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
cfg = ncfg.bert_base()
m = NBestSTCModel(cfg, labels, device="cuda", compute_dtype=torch.bfloat16, dropout=0.3, seed=999)
m.load_reference_state(synth.model_state(cfg, labels, seed=999))
m.train()
optim = HipBertAdam(m, lr=3e-5, bert_lr=3e-5, warmup=0.1, t_total=100000)
vocab = json.load(open(os.path.join(ROOT, "tests", "golden", "text_vocab.json")))
z = np.load(os.path.join(ROOT, "tests", "golden", "case_text.npz"))
memory = dict(label2idx=json.loads(str(z["label2idx"])), idx2label=labels.idx2label)
data = trainer.read_wcn_data(os.path.join(ROOT, "tests", "golden", "valid_512.txt"))
reps = (n_utts + len(data[0]) - 1) // len(data[0])
data = tuple(list(x) * reps for x in data)
opt = types.SimpleNamespace(batchSize=256, tokenizer=inputs.WordPieceTokenizer(vocab), pre_trained_model="bert", tod_pre_trained_model=None,
                            without_system_act=False, add_l2_loss=False, add_segment_ids=True, n_best=5, max_seq_len=None, random_seed=999,
                            optimizer=optim)
split = trainer.EncodedSplit(data, opt, memory)
trainer.train_epoch(m, split, opt, memory, epoch=0)          # warm the shape cache
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
trainer.train_epoch(m, split, opt, memory, epoch=1)
pr.disable()
torch.cuda.synchronize()
dt = time.time() - t0
steps = (len(split) + 255) // 256
print("epoch: %.2f s, %d steps, %.1f ms per step (wall)" % (dt, steps, dt / steps * 1e3))
pstats.Stats(pr).sort_stats("cumtime").print_stats(28)

# per-step timeline: host time to enqueue a step vs GPU time of the step (events), a third epoch
import statistics
lists = trainer.batch_indices(len(split), 256, shuffle=True, seed=5)
host, gpu = [], []
pf = trainer.Prefetcher(split, lists, m.device, 0, 1)
torch.cuda.synchronize()
t_loop = time.time()
for bi, mine, b in pf:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.time()
    out = trainer.train_step(m, optim, b, add_l2_loss=False, add_segment_ids=True, global_batch=len(lists[bi]))
    host.append((time.time() - t0) * 1e3)
    e1.record()
    gpu.append((e0, e1))
torch.cuda.synchronize()
print("train_step only loop: %.1f ms per step (wall)" % ((time.time() - t_loop) * 1e3 / len(lists)))
g = [a.elapsed_time(b_) for a, b_ in gpu]
print("train_step only, no metrics: host enqueue %.1f ms median (max %.1f); GPU step %.1f ms median (max %.1f)" % (
    statistics.median(host), max(host), statistics.median(g), max(g)))
print("allocator: %s" % {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "allocation.all.allocated")})
