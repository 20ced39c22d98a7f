import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import nbest_amd
from nbest_amd import hipabi as hb
M, N, K = 32768, 3072, 768
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()
A, W = r(M, K), r(N, K)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
dbg = torch.zeros(4096, dtype=torch.int64, device="cuda")
for _ in range(3):
    hb.gemm(A, W, M, N, K, out=out, U=dbg.view(torch.bfloat16))
torch.cuda.synchronize()
t = dbg.cpu().numpy()[:64].astype(np.int64)
t = t[t > 0]
print("stamps:", len(t))
d = np.diff(t)
print("entry->first barrier:", d[0])
for i in range((len(t) - 1) // 3):
    print("tile %d: main loop %6d | epilogue %6d | to next tile's first barrier %6d" % (i, d[1 + 3 * i], d[2 + 3 * i], d[3 + 3 * i] if 3 + 3 * i < len(d) else -1))
print("total", t[-1] - t[0])

allb = dbg.cpu().numpy()[1000:1000 + 512].astype(np.int64).reshape(256, 2)
st, en = allb[:, 0], allb[:, 1]
print("all workgroups: first start %d, last start +%d, first end +%d, last end +%d ticks; durations min %d median %d max %d" % (
    0, st.max() - st.min(), en.min() - st.min(), en.max() - st.min(), (en - st).min(), int(np.median(en - st)), (en - st).max()))
slow = np.argsort(en - st)[-8:]
print("slowest workgroups:", [(int(b), int((en - st)[b])) for b in slow])

