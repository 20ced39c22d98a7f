#!/usr/bin/env python3
"""Data-parallel step rehearsal on ONE GPU: N ranks share cuda:0, gradients go through gloo (the RCCL path needs one GPU
per rank).  Checks that N ranks on a sharded batch end with the parameters a single process gets on the whole batch.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dp_rehearsal.py
"""
import os
import sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, synth
from nbest_amd.model import NBestSTCModel
from nbest_amd.optim import HipBertAdam
from nbest_amd.trainer import GradReducer, broadcast_parameters, shard_bounds, train_step

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
cfg = ncfg.bert_base(num_hidden_layers=4, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
B, S, STEPS = 12, 48, 3


def build(seed):
    m = NBestSTCModel(cfg, labels, device="cuda:0", compute_dtype=torch.float32, dropout=0.0)
    m.load_reference_state(synth.model_state(cfg, labels, seed=seed))
    m.train()
    return m, HipBertAdam(m, lr=1e-3, bert_lr=1e-3, warmup=0.1, t_total=10)


batches = []
for s in range(STEPS):
    b = synth.nbest_batch(cfg, labels, B, S, n_best=5, seed=100 + s, ragged=True, trans_len=16)
    batches.append({k: torch.from_numpy(v).cuda() for k, v in b.items()})

# data parallel: every rank starts from a DIFFERENT seed; the broadcast makes rank 0's parameters win
m, opt = build(seed=5 + rank)
broadcast_parameters(m)
red = GradReducer(m.arena, n_chunks=2)
lo, hi = shard_bounds(B, rank, world)
shard = lambda b: {k: v[lo:hi].contiguous() for k, v in b.items()}
# (1) gradients of one batch: reduced shards == whole batch (tight: fp32 summation order only)
mine = shard(batches[0])
m.forward_backward(mine["ids"], mine["labels"], seg_ids=mine["seg"], trans_input_ids=mine["tids"], trans_seg_ids=mine["tseg"],
                   add_l2_loss=True, mse_grad_scale=1.0 / world, chunks=red.chunks, on_chunk_done=red.layers_ready)
red.wait()
torch.cuda.synchronize()
g_dp = m.arena.g.clone()
# (2) a few optimisation steps
for b in batches:
    train_step(m, opt, shard(b), add_l2_loss=True, add_segment_ids=True, reducer=red)
torch.cuda.synchronize()
got = m.arena.p.clone()

# single process on the whole batch (every rank computes it; rank 0 reports)
ms, opts = build(seed=5)
b0 = batches[0]
ms.forward_backward(b0["ids"], b0["labels"], seg_ids=b0["seg"], trans_input_ids=b0["tids"], trans_seg_ids=b0["tseg"], add_l2_loss=True)
torch.cuda.synchronize()
g_one = ms.arena.g.clone()
for b in batches:
    train_step(ms, opts, b, add_l2_loss=True, add_segment_ids=True)
torch.cuda.synchronize()
want = ms.arena.p
# the key-bias gradient is mathematically zero (softmax is shift invariant): what reaches it is rounding noise of
# ~1e-7, which Adam's normalisation turns into visible, order-dependent updates - in the reference too.  Compare the rest.
a = m.arena
err, err_kb = 0.0, 0.0
per = []
for sl in a.slots:
    d = (a.view(got, sl.name) - a.view(want, sl.name)).abs().max().item()
    per.append((d, sl.name))
    if sl.name.endswith("attention.self.key.bias"):
        err_kb = max(err_kb, d)
    else:
        err = max(err, d)
gerr = 0.0
for sl in a.slots:
    if sl.name.endswith("attention.self.key.bias") or "pooler" in sl.name:
        continue
    x, y = a.view(g_dp, sl.name), a.view(g_one, sl.name)
    gerr = max(gerr, (x - y).abs().max().item() / (y.abs().max().item() + 1e-30))
mean_err = (got - want).abs().mean().item()
allp = [torch.zeros_like(got) for _ in range(world)]
dist.all_gather(allp, got)
same = all(torch.equal(allp[0], x) for x in allp)
if rank == 0:
    print("world %d: reduced-shard gradient vs whole-batch gradient: max relative difference %.2e" % (world, gerr))
    print("world %d: parameters after %d BertAdam steps: max |DP - single| %.2e, mean %.2e (key biases %.2e); replicas bit-identical: %s" % (
        world, STEPS, err, mean_err, err_kb, same))
    # Adam divides by sqrt(v) + 1e-6: elements whose gradient is a near-cancellation (|g| << 1e-6 with 1e-6 of rounding
    # noise) move by an order-dependent ~1e-4 per step at lr 1e-3 - in the reference as well - hence the loose maximum
    assert gerr < 2e-5 and same and err < 1e-3 and mean_err < 2e-6, "data-parallel step does not reproduce the single-process step"
    print("OK")
dist.destroy_process_group()
