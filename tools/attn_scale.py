#!/usr/bin/env python3
"""attention backward (and forward) time against the batch size: µs per launch and ns per (sample, head) workgroup.  If the kernel were bound by
its own instruction stream the per-workgroup cost would not depend on whether the operands come from HBM (B = 256: 350 MB per launch) or from
the caches (B = 8 .. 32, re-run back to back).  python tools/attn_scale.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

S, heads, H = 128, 12, 768
for B in (8, 16, 32, 43, 64, 128, 256):
    M = B * S
    r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
    qkv, dctx = r(M, 3 * H), r(M, H)
    mask = torch.ones(B, S, dtype=torch.uint8, device="cuda")
    dbias = torch.zeros(3 * H, device="cuda")
    ctx, lse = hb.attention_fwd(qkv, mask, B, S, heads, 0.1, 3, 2)
    for _ in range(3):
        hb.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, heads, 0.1, 3, 2, dbias)
    res = {}
    for name, f in (("fwd", lambda: hb.attention_fwd(qkv, mask, B, S, heads, 0.1, 3, 2)),
                    ("bwd", lambda: hb.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, heads, 0.1, 3, 2, dbias))):
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
        res[name] = v[len(v) // 2]
    wg = B * heads
    rounds = wg / 512.0
    print("B %3d (%4d workgroups = %.2f rounds of 2 per CU): fwd %6.1f us  bwd %6.1f us = %5.1f ns per workgroup, %.1f us per round" % (
        B, wg, rounds, res["fwd"], res["bwd"], res["bwd"] * 1e3 / wg, res["bwd"] / max(rounds, 1.0)))
