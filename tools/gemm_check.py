#!/usr/bin/env python3
"""one full-size k-contiguous GEMM of the training step against torch.matmul (fp32 accumulate on the bf16 operands): worst relative error.
For experiment builds that switch kernels by environment (NBEST_SPLIT_RINGS, NBEST_TILE, ...).  python tools/gemm_check.py [M N K]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 768, 3072)
for Mx in (M, M - 37):
    g = torch.Generator(device="cpu").manual_seed(1)
    A = (torch.randn(Mx, K, generator=g) * 0.5).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    R = (torch.randn(Mx, N, generator=g)).bfloat16().cuda()
    ref = (A.float() @ W.float().t()) + R.float()
    for packed in (False, True):
        kw = {}
        if packed:
            pw, bn = hb.pack_weight(W)
            kw = dict(B_packed=pw, b_pack_bn=bn)
        out = hb.gemm(A, W, Mx, N, K, epilogue=hb.EPI_RES, R=R, **kw)
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        print("M %d N %d K %d packed %-5s: worst error %.2e of the largest entry  %s" % (Mx, N, K, packed, err, "OK" if err < 1e-2 else "FAIL"))
        assert err < 1e-2
