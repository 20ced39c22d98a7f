#!/usr/bin/env python3
"""rms error of the bf16 attention kernels (forward ctx, backward dQ / dK / dV) against fp32 autograd on the SAME bf16
inputs, next to the error of the bf16-storage oracle's attention core (oracle/bf16sim.py) - per sequence length."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb
from oracle.bf16sim import _AttnCore, _r

B, heads = 2, 4
H = heads * 64
for S in (64, 128, 200, 256, 300, 384, 512):
    g = torch.Generator().manual_seed(S)
    qkv = _r(torch.randn(B * S, 3 * H, generator=g))
    dctx = _r(torch.randn(B * S, H, generator=g) * 0.1)
    mask = torch.ones(B, S, dtype=torch.uint8)
    mask[1, S - 9:] = 0
    x = qkv.clone().requires_grad_(True)
    q, k, v = x.reshape(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    sc = (q @ k.transpose(-1, -2)) / 8.0
    sc = sc.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, H)
    ref.backward(dctx)
    gref = x.grad.reshape(B * S, 3, H)
    # bf16-storage oracle core
    y = qkv.clone().requires_grad_(True)
    q2, k2, v2 = y.reshape(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    o = _AttnCore.apply(q2, k2, v2, mask.bool(), 1.0 / 8.0).permute(0, 2, 1, 3).reshape(B * S, H)
    o.backward(dctx)
    gsim = y.grad.reshape(B * S, 3, H)
    # HIP
    qd, dd, md = qkv.bfloat16().cuda(), dctx.bfloat16().cuda(), mask.cuda()
    ctx, lse = hb.attention_fwd(qd, md, B, S, heads)
    dq = hb.attention_bwd(qd, md, ctx, dd, lse, B, S, heads).float().cpu().reshape(B * S, 3, H)
    rms = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    print("S=%3d  ctx: hip %.2e sim %.2e | dQ: hip %.2e sim %.2e | dK: hip %.2e sim %.2e | dV: hip %.2e sim %.2e" % (
        S, rms(ctx.float().cpu(), ref.detach()), rms(_r(o.detach()), ref.detach()),
        rms(dq[:, 0], gref[:, 0]), rms(_r(gsim[:, 0]), gref[:, 0]), rms(dq[:, 1], gref[:, 1]), rms(_r(gsim[:, 1]), gref[:, 1]),
        rms(dq[:, 2], gref[:, 2]), rms(_r(gsim[:, 2]), gref[:, 2])))
