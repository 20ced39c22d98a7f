#!/usr/bin/env python3
"""the 4 weight-gradient GEMM launches of one layer (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb
M, H, F = 32768, 768, 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
x, big = r(M, H), r(M, F)
shapes = [(3 * H, H, r(M, 3 * H), x), (H, H, r(M, H), x), (F, H, big, x), (H, F, r(M, H), big)]
outs = [torch.empty(n, k, dtype=torch.float32, device="cuda") for n, k, _, _ in shapes]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for (n, k, dy, a), o in zip(shapes, outs):
        hb.gemm(dy, a, n, k, M, 1, 1, hb.EPI_F32_SPLITK, out=o)
torch.cuda.synchronize()
