#!/usr/bin/env python3
"""GEMM micro-benchmark on the layer shapes of BASELINE configs[1] (M = 256*128 tokens): HIP-event
timing of nbest_gemm through the C-ABI, TFLOP/s per shape.  Usage: python tools/gemm_bench.py [--M 32768]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=32768)
    ap.add_argument("--H", type=int, default=768)
    ap.add_argument("--F", type=int, default=3072)
    ap.add_argument("--big", action="store_true", help="also 4096^3 / 8192^3 calibration shapes")
    a = ap.parse_args()
    M, H, F = a.M, a.H, a.F
    dev = "cuda"
    bf = torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)
    rows = []

    def run(name, fn, flops):
        ms = timeit(fn)
        rows.append((name, ms, flops / ms / 1e9))
        print("%-44s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flops / ms / 1e9), flush=True)

    if a.big:
        for n in (4096, 8192):
            A, B = r(n, n), r(n, n)
            out = torch.empty(n, n, dtype=bf, device=dev)
            run("NT %d^3 none" % n, lambda: hb.gemm(A, B, n, n, n, out=out), 2.0 * n ** 3)
            run("NN %d^3 none" % n, lambda: hb.gemm(A, B, n, n, n, 0, 1, out=out), 2.0 * n ** 3)
            o32 = torch.empty(n, n, dtype=torch.float32, device=dev)
            run("TN %d^3 f32 splitk" % n, lambda: hb.gemm(A, B, n, n, n, 1, 1, hb.EPI_F32_SPLITK, out=o32), 2.0 * n ** 3)
    x, hact = r(M, H), r(M, F)
    for (nm, N, K, A) in (("qkv", 3 * H, H, x), ("attn_out", H, H, x), ("ffn_up", F, H, x), ("ffn_down", H, F, hact)):
        W = r(N, K)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=bf, device=dev)
        U = torch.empty(M, N, dtype=torch.uint8, device=dev)      # gelu' in 8-bit fixed point (bf16 path)
        R = r(M, N)
        fl = 2.0 * M * N * K
        run("fwd  %-9s N=%4d K=%4d none" % (nm, N, K), lambda: hb.gemm(A, W, M, N, K, out=out), fl)
        run("fwd  %-9s N=%4d K=%4d bias" % (nm, N, K), lambda: hb.gemm(A, W, M, N, K, epilogue=hb.EPI_BIAS, bias=bias, out=out), fl)
        if nm == "ffn_up":
            run("fwd  %-9s N=%4d K=%4d bias_gelu" % (nm, N, K),
                lambda: hb.gemm(A, W, M, N, K, epilogue=hb.EPI_BIAS_GELU, bias=bias, out=out, U=U), fl)
        if nm in ("attn_out", "ffn_down"):
            run("fwd  %-9s N=%4d K=%4d bias_drop_res" % (nm, N, K),
                lambda: hb.gemm(A, W, M, N, K, epilogue=hb.EPI_BIAS_DROP_RES, bias=bias, R=R, out=out, drop_p=0.1, seed=1), fl)
        dY = r(M, N)
        dX = torch.empty(M, K, dtype=bf, device=dev)
        run("dgrad %-8s N=%4d K=%4d none" % (nm, K, N), lambda: hb.gemm(dY, W, M, K, N, 0, 1, out=dX), fl)
        dW = torch.empty(N, K, dtype=torch.float32, device=dev)
        run("wgrad %-8s %4dx%4d K=M splitk" % (nm, N, K), lambda: hb.gemm(dY, A, N, K, M, 1, 1, hb.EPI_F32_SPLITK, out=dW), fl)
    tot_ms = sum(m for n, m, t in rows if not n.startswith(("NT", "NN", "TN")))
    print("sum of listed layer-shape launches: %.3f ms" % tot_ms)


if __name__ == "__main__":
    main()
