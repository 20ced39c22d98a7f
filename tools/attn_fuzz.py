#!/usr/bin/env python3
"""attention backward over EVERY sequence length 97 .. 128 (the range of attn_bwd3_bf16_kernel), ragged key masks, dropout from the
forward's keep words: writes the results to a file; run once with the shipped library and once with a diag build under NBEST_ATTN_BWD=2
(the second structure) and compare:  python tools/attn_fuzz.py out_a.pt ; NBEST_LIB=... NBEST_ATTN_BWD=2 python tools/attn_fuzz.py out_b.pt ;
python tools/attn_fuzz.py out_a.pt out_b.pt"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    worst = 0.0
    for k in a:
        x, y = a[k].float(), b[k].float()
        err = (x - y).abs().max().item() / max(y.abs().max().item(), 1e-9)
        worst = max(worst, err)
        assert torch.isfinite(x).all() and err <= 2e-2, (k, err)
    print("attention backward, S = 97 .. 128: %d tensors, worst difference between the two structures %.2e of the tensor's largest entry" % (len(a), worst))
    sys.exit(0)

import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb

B, heads, H = 6, 12, 768
out = {}
for S in range(97, 129):
    g = torch.Generator(device="cpu").manual_seed(S)
    M = B * S
    qkv = (torch.randn(M, 3 * H, generator=g) * 0.7).bfloat16().cuda()
    dctx = (torch.randn(M, H, generator=g) * 0.5).bfloat16().cuda()
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).cuda()
    for p in (0.0, 0.1):
        keep = None
        if p > 0:
            ctx, lse, keep = hb.attention_fwd(qkv, mask, B, S, heads, p, 5, 2, want_keep=True)
        else:
            ctx, lse = hb.attention_fwd(qkv, mask, B, S, heads, p, 5, 2)
        dbias = torch.zeros(3 * H, device="cuda")
        dqkv = hb.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, heads, p, 5, 2, dbias, keep=keep)
        torch.cuda.synchronize()
        out["S%d p%.1f dqkv" % (S, p)] = dqkv.cpu()
        out["S%d p%.1f dbias" % (S, p)] = dbias.cpu()
torch.save(out, sys.argv[1])
print("wrote %d tensors" % len(out))
