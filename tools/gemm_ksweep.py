#!/usr/bin/env python3
"""K sweep of one k-contiguous GEMM shape (M x N fixed): separates the per-tile fixed cost from the per-stage cost.
    NBEST_LIB=... [NBEST_SYM=1] python tools/gemm_ksweep.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb
M = 32768
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
for K in (128, 256, 768, 1536, 3072):
    A, W = r(M, K), r(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fn = hb.gemm_prepared(A, W, M, N, K, out)
    for _ in range(5):
        fn()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3 * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    print("K %5d  %8.1f us  %7.0f TFLOP/s" % (K, t, 2.0 * M * N * K / t / 1e6), flush=True)
