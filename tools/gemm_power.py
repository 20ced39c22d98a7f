#!/usr/bin/env python3
"""Clock and board power while ONE GEMM shape runs back to back for a few seconds (rocm-smi sampled from a thread): is a kernel
at the power limit, i.e. would hiding its epilogue behind its main loop buy time or only lower the clock?
python tools/gemm_power.py [seconds]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
M, H, F = 32768, 768, 3072
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5)


def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            break
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
        pw = re.search(r"Power \(W\): ([0-9.]+)", t)
        if sclk and pw:
            out.append((int(sclk.group(1)), float(pw.group(1))))
        time.sleep(0.3)


def run(name, fn, flops):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < SEC:
        for _ in range(200):
            fn()
        n += 200
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    out = out[1:] or out
    sc = sum(o[0] for o in out) / max(len(out), 1)
    pw = sum(o[1] for o in out) / max(len(out), 1)
    print("%-34s %7.1f us %6.0f TFLOP/s   sclk %5.0f MHz   power %6.0f W   (%d samples)" % (name, us, flops / us / 1e6, sc, pw, len(out)), flush=True)


for nm, N, K, epi in (("fp8 FFN-down fwd (drop+res)", H, F, hb.EPI_BIAS_DROP_RES), ("fp8 FFN-up fwd (GELU)", F, H, hb.EPI_BIAS_GELU), ("fp8 QKV fwd (bias)", 3 * H, H, hb.EPI_BIAS)):
    A8 = r(M, K).to(torch.float8_e4m3fn).view(torch.uint8)
    W8 = r(N, K).to(torch.float8_e4m3fn).view(torch.uint8)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(R=r(M, N).bfloat16(), drop_p=0.1, seed=1) if epi == hb.EPI_BIAS_DROP_RES else {}
    run(nm, lambda: hb.gemm_fp8(A8, W8, M, N, K, bias, 1.0, epilogue=epi, out=out, **kw), 2.0 * M * N * K)
for nm, N, K, epi in (("bf16 FFN-down fwd (drop+res)", H, F, hb.EPI_BIAS_DROP_RES), ("bf16 FFN-up fwd (GELU)", F, H, hb.EPI_BIAS_GELU), ("bf16 QKV fwd (bias)", 3 * H, H, hb.EPI_BIAS)):
    A, W = r(M, K).bfloat16(), r(N, K).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    U = torch.empty(M, N, dtype=torch.uint8, device=dev)
    kw = dict(R=r(M, N).bfloat16(), drop_p=0.1, seed=1) if epi == hb.EPI_BIAS_DROP_RES else {}
    run(nm, lambda: hb.gemm(A, W, M, N, K, epilogue=epi, bias=bias, out=out, U=U if epi == hb.EPI_BIAS_GELU else None, **kw), 2.0 * M * N * K)
dY8 = r(M, F).to(torch.float8_e4m3fn).view(torch.uint8); X8 = r(M, H).to(torch.float8_e4m3fn).view(torch.uint8)
o32 = torch.empty(F, H, dtype=torch.float32, device=dev)
run("fp8 weight gradient 3072x768", lambda: hb.wgrad_fp8(dY8, X8, F, H, M, out=o32), 2.0 * M * F * H)
dY, X = r(M, F).bfloat16(), r(M, H).bfloat16()
run("bf16 weight gradient 3072x768", lambda: hb.gemm(dY, X, F, H, M, 1, 1, hb.EPI_F32_SPLITK, out=o32), 2.0 * M * F * H)
x = r(M, H).bfloat16()
g, b_ = torch.ones(H, device=dev), torch.zeros(H, device=dev)
run("LayerNorm forward (HBM-bound)", lambda: hb.layernorm_fwd(x, g, b_, 1e-12), 1.0)
