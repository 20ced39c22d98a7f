#!/usr/bin/env python3
"""correctness + timing of the persistent ping-pong GEMM (experiment build: make -C .../csrc diag; NBEST_LIB=.../diag/libnbest_diag.so NBEST_PERSISTENT=1) against the per-tile kernel and torch"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

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
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K) in [(8192 + 40, 3072, 768), (32768, 2304, 768), (32768, 3072, 768), (4096 + 256, 3072, 3072)]:
    A, W = r(M, K), (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    ref = A.float() @ W.float().t()
    for epi, name in ((hb.EPI_NONE, "none"), (hb.EPI_BIAS, "bias"), (hb.EPI_BIAS_GELU, "gelu")):
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        U = torch.empty(M, N, dtype=torch.uint8, device=dev) if epi == hb.EPI_BIAS_GELU else None
        kw = dict(out=out)
        if epi != hb.EPI_NONE:
            kw["bias"] = bias
        if U is not None:
            kw["U"] = U
        hb.gemm(A, W, M, N, K, False, False, epi, **kw)
        torch.cuda.synchronize()
        want = ref if epi == hb.EPI_NONE else ref + bias
        if epi == hb.EPI_BIAS_GELU:
            u = want
            want = torch.nn.functional.gelu(u)
            cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
            gp = cdf + u * torch.exp(-0.5 * u * u) / (2 * 3.141592653589793) ** 0.5
            eu = (hb.gelu_d_decode(U) - gp).abs().max().item()
        else:
            eu = 0.0
        err = (out.float() - want).abs().max().item()
        t = timeit(lambda: hb.gemm(A, W, M, N, K, False, False, epi, **kw))
        print("M=%5d N=%4d K=%4d %-5s max|err| %.4f  gelu' err %.4f  %7.1f us  %6.0f TF/s" % (M, N, K, name, err, eu, t, 2.0 * M * N * K / t / 1e6), flush=True)
        assert err < 0.08 and eu < 0.02, "MISMATCH"
