#!/usr/bin/env python3
"""Headline benchmark: fine-tuning throughput (utterances/s) of the HIP hot path on synthetic n-best
sequences - BASELINE.json configs[1]: bert-base-uncased shape, bf16, n_best=5, seq_len=128, batch 256 per
GPU; one process per GPU (torchrun), data-parallel over RCCL, weak scaling.

A step = forward (embeddings, 12 layers, STC heads + losses) + backward + gradient all-reduce +
BertAdam (per-tensor clip) with dropout ON (hidden 0.1, attention 0.1, heads 0.3, the shipped script's
values).  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: this process (which has imported nothing
    that could touch a GPU) starts one child per GPU through torch.distributed.run on 127.0.0.1, relays their output
    (rank 0 prints the JSON line) and exits with the launcher's code.  Nothing is exec'd."""
    if "RANK" in os.environ:
        return
    n = 1
    for i, tok in enumerate(sys.argv):
        if tok == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif tok.startswith("--gpus="):
            n = int(tok.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    _self_launch()

import numpy as np
import torch
import torch.distributed as dist

import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, synth, hipabi as hb
from nbest_amd.model import NBestSTCModel
from nbest_amd.optim import HipBertAdam
from nbest_amd.trainer import GradReducer, broadcast_parameters, init_distributed, train_step

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, MI355X (MI355X_MICROARCH.md)
# HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
# KiB -> bytes), measured on configs[1]; see profiles/README.md.  None until measured.
TRAFFIC_BYTES_PER_LAUNCH = 332.8e6   # (2 x 134 460 KiB FETCH_SIZE + 56 081 KiB WRITE_SIZE) x 1024, avg of the layer's 4 launches


def flops_per_utt(cfg, S, St=0):
    """3 x L*S*(24 H^2 + 4 S H) (+ the transcript pass when --add_l2_loss)"""
    H, L = cfg.hidden_size, cfg.num_hidden_layers
    f = lambda s: L * s * (24 * H * H + 4 * s * H)
    return 3.0 * (f(S) + (f(St) if St else 0))


def cpu_baseline(labels, seconds_budget=25.0):
    """the oracle (CPU restatement, kind "port") timed on this box's host cores at BASELINE configs[0]'s
    shape: bert-base fp32, B=8, S=128, n_best=5: forward + losses + backward + BertAdam."""
    from oracle.encoder import EncoderConfig as OCfg
    from oracle.model import OracleModel
    from oracle.bertadam import OracleBertAdam
    from oracle.step import train_step as oracle_step
    from oracle import stc
    # the GPU box reports the host's cores but grants a 16-core share per GPU: more threads than that thrash
    cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16))
    torch.set_num_threads(cores)
    cfg = ncfg.bert_base()
    ocfg = OCfg(**{k: v for k, v in cfg.to_dict().items() if k in OCfg.__dataclass_fields__})
    torch.manual_seed(999)
    om = OracleModel(ocfg, labels.top2bottom, labels.n_bottom, 0.3)
    om.train()
    b = synth.nbest_batch(cfg, labels, 8, 128, n_best=5, seed=999)
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    opt = OracleBertAdam(list(om.named_parameters()), lr=3e-5, bert_lr=3e-5, warmup=0.1, t_total=1000)
    b2t = stc.bottom2top_matrix(labels.top2bottom)
    oracle_step(om, opt, t, labels.top2bottom, b2t, skip_unused_transcript=True)      # warm-up
    n, t0 = 0, time.time()
    while n < 3 or (time.time() - t0 < seconds_budget * 0.5 and n < 12):
        oracle_step(om, opt, t, labels.top2bottom, b2t, skip_unused_transcript=True)
        n += 1
    dt = time.time() - t0
    return dict(value=round(8 * n / dt, 3), unit="utterances/s", cores=torch.get_num_threads(), kind="port",
                sample="%d steps of bert-base fp32 B=8 S=128 n_best=5 (fwd+loss+bwd+BertAdam), oracle on CPU" % n)


def time_dominant_kernel(M, H, F, iters=10):
    """The kernel with the largest share of the step (rocprof: profiles/): the weight-gradient GEMM
    (dW = dY^T . X over the M = B*S token rows, split-K, fp32 out; the 256x256 ping-pong kernel
    gemm2_kernel<...,true,true,F32_SPLITK> for the QKV / FFN gradients, gemm_bf16_kernel<true,true,F32_SPLITK> for the
    768x768 attention-output gradient).
    It is launched four times per layer (QKV, attention-out, FFN-up, FFN-down); this times that set on the
    launch stream with HIP events and returns (avg ms per launch, avg algorithmic flops per launch,
    avg algorithmic bytes per launch = both bf16 operands read once + the fp32 gradient written once)."""
    dev = "cuda"
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
    x, big = r(M, H), r(M, F)
    shapes = [(3 * H, H, r(M, 3 * H), x), (H, H, r(M, H), x), (F, H, big, x), (H, F, r(M, H), big)]
    outs = [torch.empty(n, k, dtype=torch.float32, device=dev) for n, k, _, _ in shapes]

    launchers = [hb.gemm_prepared(dy, a, n, k, M, o, True, True, hb.EPI_F32_SPLITK, defer_reduce=True)   # the GEMM kernel alone
                 for (n, k, dy, a), o in zip(shapes, outs)]

    # each shape is launched `iters` times in a row: like in the training step, where dY was written by the kernel just
    # before, the operands are then (partly) resident in the 256 MB Infinity Cache.  Cycling through the four shapes
    # (650 MB of operands) instead measures cold HBM reads and disagrees with the in-step rocprof average by 25 %.
    for f in launchers:
        f()
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for f in launchers:
        for _ in range(iters):
            f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (iters * len(shapes))
    flops = sum(2.0 * M * n * k for n, k, _, _ in shapes) / len(shapes)
    byts = sum(2.0 * M * (n + k) + 4.0 * n * k for n, k, _, _ in shapes) / len(shapes)
    return ms, flops, byts


def note(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--seq_len", type=int, default=128)
    ap.add_argument("--n_best", type=int, default=5)
    ap.add_argument("--model", default="bert", choices=["bert", "xlm-roberta", "xlm-roberta-large"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--add_l2_loss", action="store_true")
    ap.add_argument("--no_dropout", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    a = ap.parse_args()

    rank, world, local = init_distributed()
    assert world == a.gpus, "launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    labels = ncfg.LabelSpace.from_json(os.path.join(ROOT, "tests", "golden", "label_space.json"))
    cfg = ncfg.NAMED[a.model]()
    if a.no_dropout:
        cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model = NBestSTCModel(cfg, labels, device=dev, compute_dtype=dtype, dropout=0.0 if a.no_dropout else 0.3, seed=999)
    model.load_reference_state(synth.model_state(cfg, labels, seed=999))     # random init of the named architecture
    broadcast_parameters(model)
    model.train()
    St = a.seq_len // 4 if a.add_l2_loss else 0
    b = synth.nbest_batch(cfg, labels, a.batch, a.seq_len, n_best=a.n_best, seed=999 + rank, trans_len=St or None)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    t_total = 100000
    optim = HipBertAdam(model, lr=3e-5, bert_lr=3e-5, warmup=0.1, t_total=t_total)
    distributed = dist.is_available() and dist.is_initialized()
    reducer = GradReducer(model.arena) if distributed else None

    def step():
        return train_step(model, optim, batch, add_l2_loss=a.add_l2_loss, add_segment_ids=True, reducer=reducer)

    if rank == 0:
        note("model + batch resident on %s; warm-up %d steps" % (dev, a.warmup))
    for i in range(a.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            if rank == 0:
                note("first step done")
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        note("timing %d steps" % a.steps)
    t0 = time.time()
    for _ in range(a.steps):
        out = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss = float(out["loss_parts"].sum().item())
    assert np.isfinite(loss), "non-finite loss in the timed region"
    if distributed:
        # every rank leaves the process group together, right after the timed region: rank 0's extra measurements
        # below (dominant-kernel timing) must not keep its peers waiting inside a communicator tear-down
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        note("timed region: %.3f s, %.1f utt/s" % (dt, a.batch * world * a.steps / dt))
        utt = a.batch * world * a.steps / dt
        fpu = flops_per_utt(cfg, a.seq_len, St)
        res = {
            "metric": "utterances/sec fine-tune (bert-base, seq128, n_best=5)" if a.model == "bert" else "utterances/sec fine-tune (%s)" % a.model,
            "value": round(utt, 2), "unit": "utterances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1000 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %s shape (random init), %s, synthetic n_best=%d seq_len=%d, batch %d per GPU, "
                                   "fwd+losses+bwd+allreduce+BertAdam, dropout %s" % (
                                       {"bert": "bert-base-uncased", "xlm-roberta": "xlm-roberta-base"}.get(a.model, a.model), a.dtype, a.n_best, a.seq_len,
                                       a.batch, "off" if a.no_dropout else "on (0.1/0.1/0.3)"),
                       "global_batch": a.batch * world, "seq_len": a.seq_len, "n_best": a.n_best, "parallelism": "dp%d" % world,
                       "add_l2_loss": bool(a.add_l2_loss)},
            "flops_per_utterance": fpu,
            "step_mfma_frac": round(utt * fpu / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            "last_loss_per_utt": round(loss / a.batch, 4),
        }
        if a.dtype == "bf16":
            ms, fl, by = time_dominant_kernel(a.batch * a.seq_len, cfg.hidden_size, cfg.intermediate_size)
            ach = fl / (ms * 1e-3) / 1e12
            res["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": TRAFFIC_BYTES_PER_LAUNCH,
                               "kernel": "weight-gradient GEMM dW = dY^T.X (both operands token-major, transposed LDS reads, split-K, fp32 "
                                         "out; gemm2_kernel<256,256,...,true,true,F32_SPLITK> for QKV/FFN, gemm_bf16_kernel<true,true,"
                                         "F32_SPLITK> for the attention output), K = %d token rows; avg over the 4 launches of a layer: "
                                         "%dx%d, %dx%d, %dx%d, %dx%d" % (a.batch * a.seq_len, 3 * cfg.hidden_size, cfg.hidden_size,
                                                                        cfg.hidden_size, cfg.hidden_size, cfg.intermediate_size,
                                                                        cfg.hidden_size, cfg.hidden_size, cfg.intermediate_size),
                               "avg_launch_ms": round(ms, 4), "flops_per_launch": fl, "algorithmic_bytes_per_launch": by}
        if world == 1 and not a.no_cpu_baseline:
            note("cpu baseline (oracle on host cores) ...")
            res["cpu_baseline"] = cpu_baseline(labels)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
