#!/usr/bin/env python3
"""cycle stamps of the ping-pong slots of workgroup 0 (diagnostic build -DNBEST_DIAG=32):
make -C n-best-asr-transformer_amd/csrc diag DIAG=32 && NBEST_LIB=n-best-asr-transformer_amd/csrc/diag/libnbest_diag32.so python tools/slot_trace.py"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

M, N, K = 32768, 3072, int(sys.argv[1]) if len(sys.argv) > 1 else 768
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
A, W = r(M, K), r(N, K)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
dbg = torch.zeros(2 * 8192 + 64, dtype=torch.int64, device="cuda")
for _ in range(3):
    hb.gemm(A, W, M, N, K, out=out, U=dbg.view(torch.bfloat16))
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
d = raw[:2 * 8192].reshape(2, 2048, 4)
for name, o in (("workgroup 0 (first round)", 16000), ("workgroup 700 (third round)", 16008)):
    st = raw[o:o + 4].astype(np.int64)
    print("%s: entry -> first stage landed %d | main loop %d | epilogue incl. store drain %d | total %d cycles" % (name, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[3] - st[0]))
nk = K // 32
for g in range(2):
    t = d[g, :nk].astype(np.int64)
    t0 = t[0, 0]
    load_wait = t[:, 1] - t[:, 0]          # time parked at the barrier that ends the LOAD slot
    mfma = t[:, 2] - t[:, 1]               # MFMA slot (issue of 32 MFMAs)
    mfma_wait = t[:, 3] - t[:, 2]          # time parked at the barrier that ends the MFMA slot
    load = np.r_[0, t[1:, 0] - t[:-1, 3]]  # LOAD slot work (DMA issue, fragment reads, vmcnt wait)
    print("group %d: per-iteration cycles (s_memtime ticks @100 MHz x24 = 2.4 GHz?) median over %d stages" % (g, nk))
    print("   LOAD work %6.0f | wait@barrier1 %6.0f | MFMA slot %6.0f | wait@barrier2 %6.0f | iteration %6.0f" % (
        np.median(load[1:]), np.median(load_wait), np.median(mfma), np.median(mfma_wait), np.median(np.diff(t[:, 0]))))
    print("   first 8 iterations (LOAD, w1, MFMA, w2):", [tuple(int(x) for x in (load[i], load_wait[i], mfma[i], mfma_wait[i])) for i in range(min(8, nk))])
print("total ticks of the main loop, group 0:", int(d[0, nk - 1, 3] - d[0, 0, 0]))
