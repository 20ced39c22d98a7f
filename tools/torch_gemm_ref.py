#!/usr/bin/env python3
"""Calibration only (not part of the product): what the vendor library (hipBLASLt / rocBLAS through torch.mm) reaches on the
layer's GEMM shapes on this box, next to this build's kernels.  python tools/torch_gemm_ref.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

M, H, F = 32768, 768, 3072
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()


def timeit(f, iters=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


rows = []
# forward / dgrad style: C[M,N] = A[M,K] . W[N,K]^T
for name, N, K in (("qkv fwd", 3 * H, H), ("attn-out fwd", H, H), ("ffn-up fwd", F, H), ("ffn-down fwd", H, F), ("qkv dgrad", H, 3 * H)):
    a, w = r(M, K), r(N, K)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_lib = timeit(lambda: torch.mm(a, w.t(), out=c))
    f = hb.gemm_prepared(a, w, M, N, K, c, False, False, hb.EPI_NONE) if hasattr(hb, "gemm_prepared") else None
    t_own = None
    try:
        t_own = timeit(f)
    except Exception as e:           # signature differences: calibration tool only
        t_own = None
    rows.append((name, 2.0 * M * N * K, t_lib, t_own))
# weight gradients: dW[N,K] = dY[M,N]^T . X[M,K]
for name, N, K in (("wgrad qkv", 3 * H, H), ("wgrad attn-out", H, H), ("wgrad ffn-up", F, H), ("wgrad ffn-down", H, F)):
    dy, x = r(M, N), r(M, K)
    o = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    t_lib = timeit(lambda: torch.mm(dy.t(), x, out=o))
    o32 = torch.empty(N, K, dtype=torch.float32, device=dev)
    f = hb.gemm_prepared(dy, x, N, K, M, o32, True, True, hb.EPI_F32_SPLITK, defer_reduce=True)
    t_own = timeit(f)
    rows.append((name, 2.0 * M * N * K, t_lib, t_own))
for name, fl, tl, to in rows:
    print("%-16s lib %7.1f us %7.0f TF/s | own %s" % (name, tl * 1e3, fl / tl / 1e9, "-" if to is None else "%7.1f us %7.0f TF/s" % (to * 1e3, fl / to / 1e9)))
