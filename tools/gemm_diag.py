#!/usr/bin/env python3
"""ping-pong ablations: python tools/gemm_diag.py  (set NBEST_TILE=256x256 NBEST_GEMM_DIAG=<mask> outside)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
n = 4096
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
A, B = r(n, n), r(n, n)
out = torch.empty(n, n, dtype=torch.bfloat16, device="cuda"); o32 = torch.empty(n, n, dtype=torch.float32, device="cuda")
for name, fn in (("NT", hb.gemm_prepared(A, B, n, n, n, out)), ("NN", hb.gemm_prepared(A, B, n, n, n, out, False, True)),
                 ("TN", hb.gemm_prepared(A, B, n, n, n, o32, True, True, hb.EPI_F32_SPLITK))):
    ms = t(fn)
    print("diag=%s %s 4096^3: %.3f ms  %.0f TFLOP/s-equivalent" % (os.environ.get("NBEST_GEMM_DIAG", "0"), name, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
