#!/usr/bin/env python3
"""a few launches of the attention forward / backward at the bench shape (for rocprofv3 --pmc passes)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

B, S, heads, H = 256, 128, 12, 768
M = B * S
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
qkv, dctx = r(M, 3 * H), r(M, H)
mask = torch.ones(B, S, dtype=torch.uint8, device="cuda")
dbias = torch.zeros(3 * H, device="cuda")
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ctx, lse = hb.attention_fwd(qkv, mask, B, S, heads, p, 3, 2)
    hb.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, heads, p, 3, 2, dbias)
torch.cuda.synchronize()
