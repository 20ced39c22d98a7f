#!/usr/bin/env python3
"""time the HBM-bound kernels at the bench shape (B=256, S=128, H=768): us per launch and effective GB/s on the bytes each
must move.  python tools/hbm_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

B, S, heads, H = 256, 128, 12, 768
M = B * S
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()


def timeit(f, iters=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def row(name, us, mbytes):
    print("%-34s %7.1f us  %6.0f GB/s  (%d MB)" % (name, us, mbytes * 1e6 / us / 1e3, mbytes))


x, dy = r(M, H), r(M, H)
g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
y, stats = hb.layernorm_fwd(x, g, b, 1e-12)
row("ln_fwd", timeit(lambda: hb.layernorm_fwd(x, g, b, 1e-12)), 2 * M * H * 2 / 1e6)
row("ln_bwd (no dropout)", timeit(lambda: hb.layernorm_bwd(dy, x, stats, g, True, 0.0)), 3 * M * H * 2 / 1e6)
row("ln_bwd (dropout 0.1)", timeit(lambda: hb.layernorm_bwd(dy, x, stats, g, True, 0.1, 5, 1)), 4 * M * H * 2 / 1e6)
qkv = r(M, 3 * H)
mask = torch.ones(B, S, dtype=torch.uint8, device=dev)
ctx, lse = hb.attention_fwd(qkv, mask, B, S, heads, 0.1, 3, 2)
row("attention_fwd (dropout 0.1)", timeit(lambda: hb.attention_fwd(qkv, mask, B, S, heads, 0.1, 3, 2)), 4 * M * H * 2 / 1e6)
row("attention_fwd (no dropout)", timeit(lambda: hb.attention_fwd(qkv, mask, B, S, heads, 0.0)), 4 * M * H * 2 / 1e6)
dctx = r(M, H)
dbias = torch.zeros(3 * H, device=dev)
row("attention_bwd (dropout 0.1)", timeit(lambda: hb.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, heads, 0.1, 3, 2, dbias)), 8 * M * H * 2 / 1e6)
row("attention_bwd (no dropout)", timeit(lambda: hb.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, heads, 0.0, 0, 0, dbias)), 8 * M * H * 2 / 1e6)
big = r(M, 4 * H)
row("colsum [M,3072]", timeit(lambda: hb.colsum(big)), M * 4 * H * 2 / 1e6)
a = torch.empty(M * H * 2, dtype=torch.bfloat16, device=dev)
c = torch.empty_like(a)
row("torch copy 100 MB (reference)", timeit(lambda: c.copy_(a)), 2 * a.numel() * 2 / 1e6)
