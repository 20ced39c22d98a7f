#!/usr/bin/env python3
"""run ONE GEMM shape a few times (for rocprofv3 --pmc): python tools/gemm_one.py ffn_up [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb
which = sys.argv[1] if len(sys.argv) > 1 else "ffn_up"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M, H, F = 32768, 768, 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
if which == "ffn_up":
    A, W = r(M, H), r(F, H); out = torch.empty(M, F, dtype=torch.bfloat16, device="cuda")
    fn = lambda: hb.gemm(A, W, M, F, H, out=out)
elif which == "ffn_down":
    A, W = r(M, F), r(H, F); out = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    fn = lambda: hb.gemm(A, W, M, H, F, out=out)
elif which == "wgrad":
    dY, X = r(M, F), r(M, H); out = torch.empty(F, H, dtype=torch.float32, device="cuda")
    fn = lambda: hb.gemm(dY, X, F, H, M, 1, 1, hb.EPI_F32_SPLITK, out=out)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
