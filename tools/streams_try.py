#!/usr/bin/env python3
"""experiment: do the weight-gradient GEMMs overlap usefully with the dgrad chain when issued on a second stream?"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

M, H, F = 32768, 768, 3072
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
dRd, hact, dBig, x1, dqkv, x0 = r(M, H), r(M, F), r(M, F), r(M, H), r(M, 3 * H), r(M, H)
w2t, w1t, wqkvt = r(F, H), r(H, F), r(H, 3 * H)
u = r(M, F)


def chain(stream):
    with torch.cuda.stream(stream):
        o1 = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
        o2 = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
        o3 = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
        return [hb.gemm_prepared(dRd, w2t, M, F, H, o1, False, False, hb.EPI_NONE),     # FFN-down dgrad (ping-pong)
                hb.gemm_prepared(dBig, w1t, M, H, F, o2, False, False, hb.EPI_NONE),    # FFN-up dgrad (v1)
                hb.gemm_prepared(dqkv, wqkvt, M, H, 3 * H, o3, False, False, hb.EPI_NONE)]


def wgrads(stream):
    with torch.cuda.stream(stream):
        g2 = torch.empty(H, F, dtype=torch.float32, device=dev)
        g1 = torch.empty(F, H, dtype=torch.float32, device=dev)
        gq = torch.empty(3 * H, H, dtype=torch.float32, device=dev)
        return [hb.gemm_prepared(dRd, hact, H, F, M, g2, True, True, hb.EPI_F32_SPLITK),
                hb.gemm_prepared(dBig, x1, F, H, M, g1, True, True, hb.EPI_F32_SPLITK),
                hb.gemm_prepared(dqkv, x0, 3 * H, H, M, gq, True, True, hb.EPI_F32_SPLITK)]


def run(fa, fb, streams, iters=10):
    for _ in range(2):
        for f in fa + fb:
            f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for _ in range(iters):
        for a, b in zip(fa, fb):      # interleaved issue order, as the backward would
            a()
            b()
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


one = run(chain(s1), wgrads(s1), [s1])
two = run(chain(s1), wgrads(s2), [s1, s2])
print("dgrad chain + wgrads on ONE stream : %.1f us per layer-equivalent" % one)
print("dgrad chain on s1, wgrads on s2    : %.1f us" % two)
