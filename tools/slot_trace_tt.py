#!/usr/bin/env python3
"""cycle stamps of the ping-pong slots of the WEIGHT-GRADIENT kernel, workgroup 0 (diagnostic build -DNBEST_DIAG=32):
make -C n-best-asr-transformer_amd/csrc diag DIAG=32 && NBEST_LIB=n-best-asr-transformer_amd/csrc/diag/libnbest_diag32.so python tools/slot_trace_tt.py
Per group (one wave of each) and period (= one K stage): the two halves (group 0: LOAD | MFMA, group 1: MFMA | LOAD) and the barrier."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

K, Mo, No = 32768, 768, 3072        # tokens, output rows / columns (FFN-down weight gradient)
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
dY, X = r(K, Mo), r(K, No)
out = torch.empty(Mo, No, dtype=torch.float32, device="cuda")
dbg = torch.zeros(2 * 8192 + 64, dtype=torch.int64, device="cuda")
for _ in range(3):
    hb.gemm(dY, X, Mo, No, K, 1, 1, hb.EPI_F32_SPLITK, out=out, U=dbg.view(torch.uint8))
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
d = raw[:2 * 8192].reshape(2, 2048, 4)
nk = int((d[0, :, 0] != 0).sum())
print("stages stamped: %d" % nk)
for g in range(2):
    t = d[g, :nk].astype(np.int64)
    first, second, bar = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    names = ("LOAD slot", "MFMA slot") if g == 0 else ("MFMA slot", "LOAD slot")
    sel = slice(4, nk - 4)
    print("group %d, median over stages 4..%d (s_memtime ticks): %s %6.0f | %s %6.0f | barrier %6.0f | period %6.0f" % (
        g, nk - 4, names[0], np.median(first[sel]), names[1], np.median(second[sel]), np.median(bar[sel]), np.median(np.diff(t[:, 0])[sel])))
    print("   periods 20..27 (first half, second half, barrier):", [tuple(int(x) for x in (first[i], second[i], bar[i])) for i in range(20, min(28, nk))])
