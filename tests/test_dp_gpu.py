"""Data-parallel step on the GPU: N fresh processes share cuda:0 and exchange gradients over gloo (the RCCL transport
needs one GPU per rank; everything above the transport - sharding, chunked backward, bucketed SUM all-reduce of slices of
the gradient arena, split BertAdam, loss bookkeeping - is the code that runs over RCCL on an 8-GPU node).

Checked against ONE process on the whole batch, fp32, dropout off, uneven shards (5 utterances over 2 ranks = 3 + 2,
7 over 3 = 3 + 2 + 2), with the CLS-MSE term on (its B_local / B_global weighting is what uneven shards exercise):
  * all-reduced shard gradients == whole-batch gradients, <= 2e-5 of each tensor's largest entry (fp32 summation order);
  * after 3 BertAdam steps the replicas are bit-identical;
  * the loss record of the sharded epoch bookkeeping equals the single-process record;
  * parameters after 3 steps: mean |DP - single| <= 2e-6.  The maximum is NOT tightly bounded, and the log shows why: BertAdam
    divides by sqrt(v) + 1e-6, so an element whose gradient is a near-cancellation (|g| below ~1e-6, i.e. rounding noise of
    the 1e-5-relative summation-order difference) moves by an order-dependent +-lr*3.16 per step - in the reference too.
    The per-tensor table (gpurun_out/dp_equivalence.log) lists max / mean and the share of such elements.
"""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
LOG = os.path.join(ROOT, "gpurun_out", "dp_equivalence.log")


def _worker(rank, world, port, B, sparse, q, dtype_name="float32"):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import nbest_amd  # noqa: F401
        from nbest_amd import config as ncfg, synth
        from nbest_amd.model import NBestSTCModel
        from nbest_amd.optim import HipBertAdam
        from nbest_amd.trainer import GradReducer, _finish, broadcast_parameters, shard_bounds, train_step
        labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
        cfg = ncfg.bert_base(num_hidden_layers=4, vocab_size=3000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        S, STEPS, LR = 48, 3, 1e-3

        cdt = getattr(torch, dtype_name)

        def build(seed, shard=False):
            m = NBestSTCModel(cfg, labels, device="cuda:0", compute_dtype=cdt, dropout=0.0)
            m.load_reference_state(synth.model_state(cfg, labels, seed=seed))
            m.train()
            return m, HipBertAdam(m, lr=LR, bert_lr=LR, warmup=0.1, t_total=10, shard=shard)

        batches = []
        for s in range(STEPS):
            b = synth.nbest_batch(cfg, labels, B, S, n_best=5, seed=100 + s, ragged=True, trans_len=16)
            batches.append({k: torch.from_numpy(v).cuda() for k, v in b.items()})
        lo, hi = shard_bounds(B, rank, world)
        shard = lambda b: {k: v[lo:hi].contiguous() for k, v in b.items()}

        # data parallel: every rank starts from a DIFFERENT seed; the broadcast makes rank 0's parameters win
        m, opt = build(seed=5 + rank)
        broadcast_parameters(m)
        red = GradReducer(m.arena, n_chunks=2, sparse_word_grad=sparse)
        red.set_step_tokens(shard(batches[0])["ids"], shard(batches[0])["tids"])
        mine = shard(batches[0])
        m.forward_backward(mine["ids"], mine["labels"], seg_ids=mine["seg"], trans_input_ids=mine["tids"], trans_seg_ids=mine["tseg"],
                           add_l2_loss=True, mse_grad_scale=(hi - lo) / B, chunks=red.chunks, on_chunk_done=red.layers_ready)
        red.wait()
        torch.cuda.synchronize()
        g_dp = m.arena.g.clone()
        m.step_counter = 0
        losses = []
        for b in batches:
            out = train_step(m, opt, shard(b), add_l2_loss=True, add_segment_ids=True, reducer=red, global_batch=B)
            losses.append((out["loss_parts"].clone(), hi - lo, B))
        torch.cuda.synchronize()
        got = m.arena.p.clone()
        rec_dp, _, _ = _finish(losses, (0, 0, 0, 0, 0), "cpu", STEPS)

        # control: the replicated path a second time (is the step itself bit-reproducible from run to run?)
        mc, optc = build(seed=5 + rank)
        broadcast_parameters(mc)
        redc = GradReducer(mc.arena, n_chunks=2, sparse_word_grad=sparse)
        for b in batches:
            train_step(mc, optc, shard(b), add_l2_loss=True, add_segment_ids=True, reducer=redc, global_batch=B)
        torch.cuda.synchronize()
        gotc = mc.arena.p.clone()

        # the SHARDED optimizer (gradients reduced to the owner of each arena range, every rank updates its own range, owners
        # broadcast the compute copy): same start, same shards, same steps
        m2, opt2 = build(seed=5 + rank, shard=True)
        assert opt2.sharded and len(opt2.owner_ranges) == world
        broadcast_parameters(m2)
        red2 = GradReducer(m2.arena, n_chunks=2, sparse_word_grad=sparse, owner_ranges=opt2.owner_ranges)
        for b in batches:
            train_step(m2, opt2, shard(b), add_l2_loss=True, add_segment_ids=True, reducer=red2, global_batch=B)
        torch.cuda.synchronize()
        ws0 = m.arena.by_name["bert_encoder.embeddings.word_embeddings.weight"]
        w_same = torch.equal(m2.arena.weights, m.arena.weights)   # what the next step would read (word table included)
        opt2.gather_master()
        torch.cuda.synchronize()
        got2 = m2.arena.p.clone()
        # (rounds 1-3 excluded the word table here: its gradient was summed with fp32 atomics.  Round 4: a segmented reduce in a
        # fixed order - bit-equality is asked of the WHOLE arena, and of the control run.)
        ws_ = m.arena.by_name["bert_encoder.embeddings.word_embeddings.weight"]
        wd = (got2 - got)[ws_.offset:ws_.offset + ws_.numel].abs()
        sh = dict(equal=torch.equal(got2, got), weights_equal=w_same, max=(got2 - got).abs().max().item(),
                  mean=(got2 - got).abs().mean().item(), word_max=wd.max().item(), word_mean=wd.mean().item(),
                  m_equal=torch.equal(m2.arena.m, m.arena.m), owned=[r_ for r_ in opt2.owner_ranges[rank]],
                  ctrl_equal=torch.equal(gotc, got), ctrl_max=(gotc - got).abs().max().item())
        g2h = got2.cpu()
        all2 = [torch.zeros_like(g2h) for _ in range(world)]
        dist.all_gather(all2, g2h)
        sh["replicas_same"] = all(torch.equal(all2[0], x) for x in all2)

        # one process on the whole batch (every rank computes it redundantly)
        ms, opts = build(seed=5)
        b0 = batches[0]
        ms.forward_backward(b0["ids"], b0["labels"], seg_ids=b0["seg"], trans_input_ids=b0["tids"], trans_seg_ids=b0["tseg"], add_l2_loss=True)
        torch.cuda.synchronize()
        g_one = ms.arena.g.clone()
        ms.step_counter = 0
        rec_one = 0.0
        for b in batches:
            out = train_step(ms, opts, b, add_l2_loss=True, add_segment_ids=True)
            rec_one += out["loss_parts"].double().sum().item() / B / STEPS
        torch.cuda.synchronize()
        want = ms.arena.p
        a = m.arena
        rows, gerr = [], 0.0
        for sl in a.slots:
            if "pooler" in sl.name:
                continue
            x, y = a.view(g_dp, sl.name), a.view(g_one, sl.name)
            # the key-bias gradient is mathematically zero (softmax is shift invariant): only rounding noise reaches it
            ge = 0.0 if sl.name.endswith("attention.self.key.bias") else (x - y).abs().max().item() / (y.abs().max().item() + 1e-30)
            gerr = max(gerr, ge)
            d = (a.view(got, sl.name) - a.view(want, sl.name)).abs()
            rows.append((sl.name, ge, d.max().item(), d.mean().item(), (d > 0.01 * LR).float().mean().item()))
        got_h = got.cpu()
        allp = [torch.zeros_like(got_h) for _ in range(world)]
        dist.all_gather(allp, got_h)
        same = all(torch.equal(allp[0], x) for x in allp)
        mean_err = (got - want).abs().mean().item()
        dist.destroy_process_group()
        q.put((rank, dict(gerr=gerr, same=same, mean_err=mean_err, max_err=(got - want).abs().max().item(), rec_dp=rec_dp,
                          rec_one=rec_one, rows=rows if rank == 0 else None, sharded=sh)))
    except BaseException as e:                      # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, dict(error=traceback.format_exc() + repr(e))))


@pytest.mark.parametrize("world,B,sparse,dtype_name", [(2, 5, False, "float32"), (3, 7, True, "float32"), (2, 6, False, "bfloat16")])
def test_dp_step_equals_single_process(world, B, sparse, dtype_name):
    """sparse: the word-embedding gradient travels as (row ids, row values) instead of through the dense all-reduce.
    Every case also runs the SHARDED optimizer (HipBertAdam(shard=True) + reduce-to-owner) next to the replicated one: its
    replicas must be bit-identical, and its parameters equal the replicated path's - to the last bit where the gradient sum has
    two terms (world 2: a + b is the same number whichever collective forms it), to summation-order noise at world 3.  The
    bfloat16 case checks exactly that on the production dtype (bf16 compute copy + fp32 small tensors broadcast by the owners)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29620 + world + (7 if dtype_name != "float32" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, sparse, q, dtype_name)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    for r in range(world):
        assert "error" not in res[r], res[r]["error"]
    r0 = res[0]
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write("world %d, batch %d (uneven shards), %s, 4 layers, lr 1e-3, 3 BertAdam steps, word-embedding gradient exchange: %s\n" % (
            world, B, dtype_name, "sparse rows" if sparse else "dense all-reduce"))
        f.write("  reduced-shard gradient vs whole-batch gradient: max relative difference %.2e (bound 2e-5)\n" % r0["gerr"])
        f.write("  parameters: mean |DP - single| %.2e (bound 2e-6), max %.2e; replicas bit-identical: %s\n" % (
            r0["mean_err"], r0["max_err"], all(res[r]["same"] for r in range(world))))
        f.write("  loss record: DP %.8f single %.8f\n" % (r0["rec_dp"], r0["rec_one"]))
        f.write("  %-64s %10s %10s %10s %s\n" % ("tensor", "grad rel", "max |dp|", "mean |dp|", "share of elements off by > 1% of lr"))
        for name, ge, mx, mean, share in sorted(r0["rows"], key=lambda t: -t[2])[:12]:
            f.write("  %-64s %10.2e %10.2e %10.2e %.2e\n" % (name[-64:], ge, mx, mean, share))
        sh = r0["sharded"]
        f.write("  sharded optimizer vs replicated (whole arena, word table included): parameters bit-equal %s (max |d| %.2e), "
                "compute copy bit-equal %s, moments bit-equal %s, replicas bit-identical %s; word table max |d| %.2e mean %.2e; rank 0 owns %s\n" % (
                    sh["equal"], sh["max"], sh["weights_equal"], sh["m_equal"], all(res[r]["sharded"]["replicas_same"] for r in range(world)),
                    sh["word_max"], sh["word_mean"], sh["owned"]))
        f.write("  control - the replicated path run twice: bit-equal %s (max |d| %.2e)\n" % (sh["ctrl_equal"], sh["ctrl_max"]))
    for r in range(world):
        sh = res[r]["sharded"]
        assert sh["replicas_same"], "sharded replicas diverged"
        assert sh["ctrl_equal"], ("the replicated data-parallel step is not bit-reproducible run to run", sh)
        if world == 2:                     # a + b is the same number whichever collective forms it
            assert sh["equal"] and sh["weights_equal"] and sh["m_equal"], ("sharded optimizer differs from the replicated one", sh)
        else:                              # world 3: reduce-to-owner and all-reduce may add the three terms in different orders
            assert sh["mean"] < 2e-6 and sh["max"] <= 1e-4, sh
        assert res[r]["same"], "replicas diverged"
        if dtype_name != "float32":
            continue                       # the single-process comparison below carries fp32 bars
        assert res[r]["gerr"] < 2e-5, res[r]["gerr"]
        assert res[r]["mean_err"] < 2e-6, res[r]["mean_err"]
        assert abs(res[r]["rec_dp"] - res[r]["rec_one"]) <= 1e-5 * abs(res[r]["rec_one"]), (res[r]["rec_dp"], res[r]["rec_one"])


@pytest.mark.parametrize("world,model,batch", [(2, "bert", 16), (3, "xlm-roberta", 8)])
def test_bench_multi_rank_code_path_rehearsal(world, model, batch):
    """`bench.py --gpus N` end to end on the one-GPU box: NBEST_BENCH_REHEARSAL=1 puts the N ranks on cuda:0 over gloo, so the
    self-launch through torch.distributed.run, the sharded optimizer, the reduce-to-owner buckets (bert) / the sparse row exchange
    (xlm-roberta), the barrier-bracketed timed region, the tear-down of the process group and rank 0's in-step kernel timing AFTER the
    tear-down (a replicated update there: no collective may be left in the step) all run - what the driver's N = 2 / 4 / 8 runs execute
    over RCCL.  Only the code path is checked; the timings of ranks sharing one GPU mean nothing."""
    import json
    import subprocess
    env = dict(os.environ, NBEST_BENCH_REHEARSAL="1")
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                          "--no_cpu_baseline", "--model", model, "--batch", str(batch)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == world and rec["value"] > 0 and rec["scaling"] == "weak" and "rehearsal" in rec
    assert rec["roofline"]["achieved"] > 0


@pytest.mark.parametrize("extra", [[], ["--model", "xlm-roberta", "--shard_optimizer", "on", "--batch", "8"]])
def test_rccl_branch_runs_at_world_one(extra):
    """the RCCL branch itself - init_process_group("nccl", device_id=...), all_gather_into_tensor, the async
    all_reduce / reduce buckets on the communicator's stream, the side-stream count copy, barrier and tear-down - executed on the
    one-GPU box: a FRESH child under torch.distributed.run with one rank and WITHOUT the gloo rehearsal switch.  (World 1 cannot
    show a wrong sum; it shows that every call the 8-GPU run makes exists, accepts these arguments on RCCL and completes.)"""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "NBEST_BENCH_REHEARSAL", "NBEST_DP_REHEARSAL"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29733", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no_cpu_baseline"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and "rehearsal" not in rec and rec["value"] > 0
    assert rec["dist_backend"] == "nccl", rec
