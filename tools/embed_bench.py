#!/usr/bin/env python3
"""The deterministic embedding backward at the bench shape (bert-base, B=256, S=128, synthetic n-best ids): all four kernels
back to back, as the step issues them (tables zeroed by a memset, touched rows overwritten).  Run under
`rocprofv3 --kernel-trace --stats` for per-kernel times; NBEST_LIB=<variant .so> for -DNBEST_EMB_TPC builds.
    python tools/embed_bench.py [B] [S]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbest_amd  # noqa: F401
from nbest_amd import config as ncfg, hipabi as hb, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, V = 768, 30522
dev = "cuda"
labels = ncfg.LabelSpace.from_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "label_space.json"))
cfg = ncfg.bert_base()
b = synth.nbest_batch(cfg, labels, B, S, n_best=5, seed=999)
ids = torch.from_numpy(b["ids"]).to(dev)
seg = torch.from_numpy(b["seg"]).to(dev)
pos = torch.arange(S, device=dev)[None, :].expand(B, S).contiguous()
perm = torch.from_numpy(np.argsort(b["ids"].ravel(), kind="stable").astype(np.int32)).to(dev)
r = lambda *s: (torch.randn(*s, device=dev) * 0.05).bfloat16()
word, tt, pt = r(V, H), r(2, H), r(512, H)
gam, bet = torch.ones(H, device=dev), torch.zeros(H, device=dev)
out, stats = hb.embed_ln_fwd(ids, seg, pos, word, tt, pt, gam, bet, 1e-12)
dout = r(B * S, H)
dword = torch.zeros(V, H, device=dev)
dtt, dpt = torch.zeros(2, H, device=dev), torch.zeros(512, H, device=dev)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
L = hb.lib()
ws = torch.empty(L.nbest_embed_bwd_ws_bytes(B * S, H), dtype=torch.uint8, device=dev)


def run():
    dword.zero_()
    hb.check(L.nbest_embed_ln_bwd(hb.ptr(ids), hb.ptr(seg), hb.ptr(pos), hb.ptr(perm), hb.ptr(word), hb.ptr(tt), hb.ptr(pt), hb.ptr(gam),
                                  hb.ptr(stats), hb.ptr(dout), hb.ptr(dword), hb.ptr(dtt), hb.ptr(dpt), hb.ptr(dg), hb.ptr(db), B, S, H, 2,
                                  hb.dtype_code(torch.bfloat16), 0, -1, 0, 0, 0.1, 5, 7, hb.ptr(ws), ws.numel(), hb.stream_ptr()), "embed_ln_bwd")


for _ in range(3):
    run()
torch.cuda.synchronize()
ref = dword.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
uniq = int(torch.unique(ids).numel())
print("%s: B=%d S=%d unique rows %d: %.1f us per call incl. the table memset; bit-reproducible: %s" % (
    os.path.basename(hb.LIB_PATH), B, S, uniq, e0.elapsed_time(e1) / 20 * 1e3, torch.equal(ref, dword)))
