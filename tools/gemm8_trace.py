#!/usr/bin/env python3
"""Timeline of the two-workgroups-per-CU fp8 GEMM (diag build only): per CU, when each workgroup ran its main loop and its
epilogue.  NBEST_LIB=.../diag/libnbest_diag.so python tools/gemm8_trace.py [stagger_10ns]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb
M, N, K = 32768, 3072, 768
os.environ["NBEST_GEMM8_WN"] = "2"
os.environ["NBEST_GEMM8_STAGGER"] = sys.argv[1] if len(sys.argv) > 1 else "0"
dev = "cuda"
A8 = (torch.randn(M, K, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
W8 = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
nwg = (M // 256) * (N // 128)
tr = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
for _ in range(3):
    hb.gemm_fp8(A8, W8, M, N, K, bias, 1.0, epilogue=hb.EPI_BIAS_GELU, out=out)
torch.cuda.synchronize()
assert hb.lib().nbest_experiment_trace8(hb.ptr(tr)) == 0
hb.gemm_fp8(A8, W8, M, N, K, bias, 1.0, epilogue=hb.EPI_BIAS_GELU, out=out)
torch.cuda.synchronize()
hb.lib().nbest_experiment_trace8(None)
t = tr.cpu().numpy().reshape(nwg, 4)
hw = t[:, 0] & 0xffffffff
xcc = (t[:, 0] >> 32) & 7
key = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
slot = hw & 15
t0 = t[:, 1].min()
print("workgroups %d, distinct CUs %d, kernel span %.1f us" % (nwg, len(set(key.tolist())), (t[:, 3].max() - t0) / 100.0))
print("main loop  mean %.2f us   epilogue mean %.2f us" % (((t[:, 2] - t[:, 1]).mean()) / 100.0, ((t[:, 3] - t[:, 2]).mean()) / 100.0))
for k in sorted(set(key.tolist()))[:3]:
    idx = np.nonzero(key == k)[0]
    idx = idx[np.argsort(t[idx, 1])]
    print("CU key %4d: %d workgroups" % (k, len(idx)))
    for i in idx:
        print("   wg %5d slot %2d  start %7.2f  main_end %7.2f  end %7.2f" % (i, slot[i], (t[i, 1] - t0) / 100.0, (t[i, 2] - t0) / 100.0, (t[i, 3] - t0) / 100.0))
