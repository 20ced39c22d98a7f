"""world_size-2 gloo test (CPU) of the data-parallel plumbing: sharding, the bucketed SUM all-reduce over
slices of the flat gradient arena in chunk order, and the metric reduction - the same GradReducer /
shard_bounds / _finish code that runs over RCCL on the GPUs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nbest_amd  # noqa: F401
    from nbest_amd import config as ncfg, trainer
    from nbest_amd.arena import ParamArena
    labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
    cfg = ncfg.bert_base(num_hidden_layers=4, vocab_size=300, hidden_size=128, num_attention_heads=2, intermediate_size=256)
    a = ParamArena(cfg, labels, "cpu", compute_dtype=torch.float32)
    torch.manual_seed(rank)
    a.g.copy_(torch.randn(a.total))
    local = a.g.clone()
    red = trainer.GradReducer(a, n_chunks=3)      # explicit: 4 layers -> (0,1) (1,3) (3,4)
    assert red.chunks == [(0, 1), (1, 3), (3, 4)]
    for lo, hi in sorted(red.chunks, reverse=True):              # the order the backward produces them
        red.layers_ready(lo, hi)
    assert len(red.pending) == 4 and len(red.pending_emb) == 1    # heads + 3 layer buckets | embedding tables (last)
    red.wait_layers()
    assert red.pending == [] and len(red.pending_emb) == 1
    red.wait()
    assert red.pending_emb == []
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered)
    covered = torch.zeros(a.total, dtype=torch.bool)
    for lo, hi in [a.emb_range, a.heads_range] + [(a.layer_range[l][0], a.layer_range[l][1]) for l in range(4)]:
        covered[lo:hi] = True
    ok = torch.allclose(a.g[covered], want[covered], atol=1e-6)
    # pooler slots are never reduced (no gradient): they must be untouched
    pl = a.by_name["bert_encoder.pooler.dense.weight"]
    ok = ok and torch.equal(a.g[pl.offset:pl.offset + pl.numel], local[pl.offset:pl.offset + pl.numel])
    # parameter broadcast + metric reduction
    a.p.fill_(float(rank + 1))
    m = type("M", (), {"arena": a})()
    trainer.broadcast_parameters(m)
    ok = ok and bool((a.p == 1.0).all())
    # uneven shards of a 5-utterance batch (3 + 2): sum-type parts add up, the MSE part (a mean over the shard's rows) is
    # weighted by B_local / B_global -> the record of the single process on the whole batch
    b_local = 3 if rank == 0 else 2
    parts = torch.tensor([2.0 * (rank + 1), 1.0, 0.5, 0.3 * (rank + 1)])
    loss, (p, r, f), acc = trainer._finish([(parts, b_local, 5)], (3 + rank, 1, 2, 1 + rank, 4), "cpu", 1)
    want_loss = ((2.0 + 1.0 + 0.5) + (4.0 + 1.0 + 0.5) + 0.3 * 3 / 5 + 0.6 * 2 / 5) / 5
    ok = ok and abs(loss - want_loss) < 1e-6 and abs(acc - 100 * 3 / 8) < 1e-9 and abs(p - 100 * 7 / 9) < 1e-9
    lo, hi = trainer.shard_bounds(7, rank, world)
    ok = ok and (lo, hi) == ((0, 4) if rank == 0 else (4, 7))
    # short final batch: rank 1 has no utterance, joins the step's collectives with zeros and ends with rank 0's sums
    red2 = trainer.GradReducer(a, n_chunks=2)
    if rank == 0:
        a.g.copy_(local)
        for lo_, hi_ in reversed(red2.chunks):
            red2.layers_ready(lo_, hi_)
        red2.wait()
    else:
        red2.contribute_nothing()
    ref0 = local.clone()
    dist.broadcast(ref0, src=0)
    for lo_, hi_ in [a.heads_range, a.emb_range] + [(a.layer_range[l][0], a.layer_range[l][1]) for l in range(len(a.layer_range))]:
        ok = ok and torch.equal(a.g[lo_:hi_], ref0[lo_:hi_])
    # sparse exchange of the word-embedding gradient (XLM-R's 250 002-row table under DP): rows touched by the rank's own
    # tokens travel as (ids, values); the rebuilt table equals the dense SUM and is bit-identical on both ranks
    red3 = trainer.GradReducer(a, n_chunks=2, sparse_word_grad=True)
    ws = a.by_name["bert_encoder.embeddings.word_embeddings.weight"]
    V, Hd = ws.shape
    gen = torch.Generator().manual_seed(100 + rank)
    toks = torch.randint(0, V, (3 + rank, 17), generator=gen)               # ragged across ranks, overlapping rows
    toks[0, :4] = torch.tensor([5, 6, 7, 5])                                # rows both ranks touch, one of them twice
    a.g.copy_(torch.randn(a.total, generator=gen))
    G = a.g[ws.offset:ws.offset + ws.numel].view(V, Hd)
    dense_rows = torch.zeros(V, dtype=torch.bool)
    dense_rows[toks.reshape(-1)] = True
    G[~dense_rows] = 0                                                      # what the embedding backward leaves: only touched rows
    mine = a.g.clone()
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    want3 = both[0] + both[1]
    red3.set_step_tokens(toks)                                              # device-side unique (no host-built row list)
    for lo_, hi_ in reversed(red3.chunks):
        red3.layers_ready(lo_, hi_)
    red3.wait()
    for lo_, hi_ in [a.emb_range, a.heads_range] + [(a.layer_range[l][0], a.layer_range[l][1]) for l in range(len(a.layer_range))]:
        ok = ok and torch.allclose(a.g[lo_:hi_], want3[lo_:hi_], atol=1e-6)
    got3 = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got3, a.g.clone())
    ok = ok and torch.equal(got3[0][:ws.numel], got3[1][:ws.numel])         # replicas bit-identical
    # the same through a host-built row list (what EncodedSplit / bench.py hand over), and a rank with no tokens at all
    a.g.copy_(mine)
    if rank == 0:
        red3.set_step_tokens(rows=torch.unique(toks.reshape(-1)))
        for lo_, hi_ in reversed(red3.chunks):
            red3.layers_ready(lo_, hi_)
        red3.wait()
        ok = ok and torch.allclose(a.g[:ws.numel], mine[:ws.numel], atol=0)   # rank 1 contributed zeros
    else:
        red3.contribute_nothing()
        ok = ok and torch.equal(a.g[:ws.numel], both[0][:ws.numel])
    # sharded optimizer: every bucket travels as a REDUCE of its intersection with each owner's arena range to that owner
    a.g.copy_(local)
    cut = a.layer_range[1][1] - 1000                        # an ownership boundary inside a layer (and inside a bucket)
    emb_cut = a.emb_range[1] // 2
    owners = [[(a.layer_range[0][0], cut), (0, emb_cut)], [(cut, a.total), (emb_cut, a.emb_range[1])]]
    red4 = trainer.GradReducer(a, n_chunks=3, owner_ranges=owners)
    for lo_, hi_ in sorted(red4.chunks, reverse=True):
        red4.layers_ready(lo_, hi_)
    red4.wait()
    for lo_, hi_ in owners[rank]:
        seg = covered[lo_:hi_]                              # (pooler slots are in no bucket)
        ok = ok and torch.allclose(a.g[lo_:hi_][seg], want[lo_:hi_][seg], atol=1e-6)
    # eval cases: every rank holds its slice of each batch; merged list = dataset order on every rank
    mine = [(b, rank, [("b%d" % b, "r%d" % rank, i) for i in range(2 - rank + b % 2)]) for b in range(3)]
    merged = trainer.merge_cases(mine)
    want = [("b%d" % b, "r%d" % r, i) for b in range(3) for r in range(2) for i in range(2 - r + b % 2)]
    ok = ok and merged == want
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_allreduce_and_metrics_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
