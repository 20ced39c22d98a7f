#!/usr/bin/env python3
"""The sixteen GEMM launches of one encoder layer (forward, dgrad, weight gradient) exactly as the training step issues them
(operand layouts, epilogues, fused column sums), timed with HIP events through the C-ABI: per launch µs / TFLOP/s, rounds
interleaved over the whole list so that every shape sees the same clock state.  BASELINE configs[1] shapes by default.
    python tools/layer_gemms.py [--M 32768] [--rounds 12] [--tag name]
With a `make diag` build (NBEST_LIB=.../libnbest_diag.so) NBEST_TILE / NBEST_GEMM force a tile / kernel generation."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa: F401
from nbest_amd import hipabi as hb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=32768)
    ap.add_argument("--H", type=int, default=768)
    ap.add_argument("--F", type=int, default=3072)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--inner", type=int, default=3)
    ap.add_argument("--tag", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--no_pack", action="store_true")
    ap.add_argument("--prep", default="none", choices=["none", "copy", "read"],
                    help="before EVERY timed launch (outside the timed events): copy = rewrite the activation operand with plain stores (an operand the "
                         "previous kernel has just written, as in the step); read = read it once (sum); none = as is (with --inner 1: last touched a "
                         "round ago)")
    a = ap.parse_args()
    M, H, F = a.M, a.H, a.F
    dev, bf = "cuda", torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)
    x, ctx, x1, hact = r(M, H), r(M, H), r(M, H), r(M, F)
    dRd, dBig, dqkv = r(M, H), r(M, F), r(M, 3 * H)
    dR = r(M, H)
    Wqkv, Wo, W1, W2 = r(3 * H, H), r(H, H), r(F, H), r(H, F)
    WqkvT, WoT, W1T, W2T = (w.t().contiguous() for w in (Wqkv, Wo, W1, W2))
    bqkv, bo, b1, b2 = (torch.randn(n, device=dev) for n in (3 * H, H, F, H))
    # the weight operands as the training step passes them: pre-packed for the kernel's tile (nbest_pack_weights); --no_pack: row by row
    PK = {}
    for nm, w in (("qkv", Wqkv), ("o", Wo), ("w1", W1), ("w2", W2), ("qkvT", WqkvT), ("oT", WoT), ("w1T", W1T), ("w2T", W2T)):
        pw, bn = (None, 0) if a.no_pack else hb.pack_weight(w)
        PK[nm] = dict(B_packed=pw, b_pack_bn=bn) if pw is not None else {}
    qkv = torch.empty(M, 3 * H, dtype=bf, device=dev)
    oH = torch.empty(M, H, dtype=bf, device=dev)
    oF = torch.empty(M, F, dtype=bf, device=dev)
    U = torch.empty(M, F, dtype=torch.uint8, device=dev)
    U.random_(0, 255)
    gb1 = torch.zeros(F, device=dev)
    g32 = {k: torch.empty(*s, dtype=torch.float32, device=dev) for k, s in
           (("qkv", (3 * H, H)), ("o", (H, H)), ("w1", (F, H)), ("w2", (H, F)))}
    E = hb
    jobs = [
        ("fwd  qkv       N2304 K768  bias", 2.0 * M * 3 * H * H, lambda: E.gemm(x, Wqkv, M, 3 * H, H, epilogue=E.EPI_BIAS, bias=bqkv, out=qkv, **PK['qkv'])),
        ("fwd  attn-out  N768  K768  bias+drop+res", 2.0 * M * H * H, lambda: E.gemm(ctx, Wo, M, H, H, epilogue=E.EPI_BIAS_DROP_RES, bias=bo, R=x, out=oH, drop_p=0.1, seed=1, drop_stream=3, **PK['o'])),
        ("fwd  ffn-up    N3072 K768  bias+gelu", 2.0 * M * F * H, lambda: E.gemm(x1, W1, M, F, H, epilogue=E.EPI_BIAS_GELU, bias=b1, U=U, out=oF, **PK['w1'])),
        ("fwd  ffn-down  N768  K3072 bias+drop+res", 2.0 * M * F * H, lambda: E.gemm(hact, W2, M, H, F, epilogue=E.EPI_BIAS_DROP_RES, bias=b2, R=x1, out=oH, drop_p=0.1, seed=1, drop_stream=4, **PK['w2'])),
        ("dgrd ffn-down  N3072 K768  x gelu' + colsum", 2.0 * M * F * H, lambda: E.gemm(dRd, W2T, M, F, H, epilogue=E.EPI_DGELU, U=U, out=oF, colsum_out=gb1, **PK['w2T'])),
        ("dgrd ffn-up    N768  K3072 + res", 2.0 * M * F * H, lambda: E.gemm(dBig, W1T, M, H, F, epilogue=E.EPI_RES, R=dR, out=oH, **PK['w1T'])),
        ("dgrd attn-out  N768  K768  none", 2.0 * M * H * H, lambda: E.gemm(dRd, WoT, M, H, H, out=oH, **PK['oT'])),
        ("dgrd qkv       N768  K2304 + res", 2.0 * M * 3 * H * H, lambda: E.gemm(dqkv, WqkvT, M, H, 3 * H, epilogue=E.EPI_RES, R=dR, out=oH, **PK['qkvT'])),
        ("wgrd ffn-down  768x3072", 2.0 * M * F * H, lambda: E.gemm(dRd, hact, H, F, M, 1, 1, E.EPI_F32_SPLITK, out=g32["w2"])),
        ("wgrd ffn-up    3072x768", 2.0 * M * F * H, lambda: E.gemm(dBig, x1, F, H, M, 1, 1, E.EPI_F32_SPLITK, out=g32["w1"])),
        ("wgrd attn-out  768x768", 2.0 * M * H * H, lambda: E.gemm(dRd, ctx, H, H, M, 1, 1, E.EPI_F32_SPLITK, out=g32["o"])),
        ("wgrd qkv       2304x768", 2.0 * M * 3 * H * H, lambda: E.gemm(dqkv, x, 3 * H, H, M, 1, 1, E.EPI_F32_SPLITK, out=g32["qkv"])),
    ]
    act = {"fwd  qkv": x, "fwd  attn-out": ctx, "fwd  ffn-up": x1, "fwd  ffn-down": hact, "dgrd ffn-down": dRd, "dgrd ffn-up": dBig,
           "dgrd attn-out": dRd, "dgrd qkv": dqkv, "wgrd ffn-down": dRd, "wgrd ffn-up": dBig, "wgrd attn-out": dRd, "wgrd qkv": dqkv}
    spare = {id(t): t.clone() for t in act.values()}
    sink = torch.zeros(1, device=dev)

    def prep(name):
        t = next(v for k, v in act.items() if name.startswith(k))
        if a.prep == "copy":
            t.copy_(spare[id(t)])
        elif a.prep == "read":
            sink.add_(t.view(torch.int16)[::1].sum().float() * 0)
    if a.only:
        jobs = [j for j in jobs if a.only in j[0]]
    for _, _, f in jobs:          # warm-up (workspace allocation, code load)
        f(); f()
    torch.cuda.synchronize()
    times = [[] for _ in jobs]
    for _ in range(a.rounds):
        for k, (nm_, _, f) in enumerate(jobs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if a.prep != "none":
                prep(nm_)
            e0.record()
            for _ in range(a.inner):
                f()
            e1.record()
            times[k].append((e0, e1))
    torch.cuda.synchronize()
    tot = 0.0
    print("# %s  prep=%s inner=%d  M=%d  lib=%s  NBEST_TILE=%s NBEST_GEMM=%s" % (a.tag, a.prep, a.inner, M, os.path.basename(hb.LIB_PATH), os.environ.get("NBEST_TILE", "-"),
                                                             os.environ.get("NBEST_GEMM", "-")))
    for (name, fl, _), ts in zip(jobs, times):
        v = sorted(e0.elapsed_time(e1) / a.inner * 1e3 for e0, e1 in ts)
        med, mn = v[len(v) // 2], v[0]
        tot += med
        print("%-46s median %7.1f us  min %7.1f us  %6.0f TFLOP/s" % (name, med, mn, fl / med / 1e6))
    print("sum of medians: %.1f us per layer" % tot)


if __name__ == "__main__":
    main()
