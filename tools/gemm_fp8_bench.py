#!/usr/bin/env python3
"""fp8 (block-scaled MFMA) forward GEMMs next to the bf16 kernels on the layer shapes: python tools/gemm_fp8_bench.py [--M 32768 --H 768 --F 3072]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nbest_amd  # noqa
from nbest_amd import hipabi as hb
from gemm_bench import timeit
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32768); ap.add_argument("--H", type=int, default=768); ap.add_argument("--F", type=int, default=3072)
a = ap.parse_args()
M, H, F = a.M, a.H, a.F
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5)
for nm, N, K, epi in (("qkv", 3 * H, H, hb.EPI_BIAS), ("attn_out", H, H, hb.EPI_BIAS_DROP_RES), ("ffn_up", F, H, hb.EPI_BIAS_GELU), ("ffn_down", H, F, hb.EPI_BIAS_DROP_RES)):
    A, W = r(M, K).bfloat16(), r(N, K).bfloat16()
    A8, W8 = A.float().to(torch.float8_e4m3fn).view(torch.uint8), W.float().to(torch.float8_e4m3fn).view(torch.uint8)
    bias = torch.randn(N, device=dev)
    R = r(M, N).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    U = torch.empty(M, N, dtype=torch.uint8, device=dev)
    fl = 2.0 * M * N * K
    kw = dict(R=R, drop_p=0.1, seed=1) if epi == hb.EPI_BIAS_DROP_RES else {}
    t16 = timeit(lambda: hb.gemm(A, W, M, N, K, epilogue=epi, bias=bias, out=out, U=U if epi == hb.EPI_BIAS_GELU else None, **kw))
    t8 = timeit(lambda: hb.gemm_fp8(A8, W8, M, N, K, bias, 1.0, epilogue=epi, out=out, **kw))
    extra = ""
    if os.environ.get("NBEST_LIB"):       # diag build: both tile variants (NBEST_GEMM8_WN)
        for wn, stg in (("4", "0"), ("2", "0"), ("2", "300"), ("2", "600"), ("2", "1000"), ("2", "1500")):
            os.environ["NBEST_GEMM8_WN"] = wn
            os.environ["NBEST_GEMM8_STAGGER"] = stg
            t = timeit(lambda: hb.gemm_fp8(A8, W8, M, N, K, bias, 1.0, epilogue=epi, out=out, **kw))
            extra += "  wn%s/%s %5.1f" % (wn, stg, t * 1e3)
        del os.environ["NBEST_GEMM8_WN"], os.environ["NBEST_GEMM8_STAGGER"]
    print("%-9s N=%4d K=%4d epi %d   bf16 %7.1f us %6.0f TF/s | fp8 %7.1f us %6.0f TF/s  (x%.2f)%s" % (nm, N, K, epi, t16 * 1e3, fl / t16 / 1e9, t8 * 1e3, fl / t8 / 1e9, t16 / t8, extra), flush=True)

# weight gradients: bf16 TT kernel vs fp8 transposed-read kernel (K = M tokens)
for nm, Mo, No in (("qkv", 3 * H, H), ("attn_out", H, H), ("ffn_up", F, H), ("ffn_down", H, F)):
    dY, X = r(M, Mo).bfloat16(), r(M, No).bfloat16()
    dY8, X8 = dY.float().to(torch.float8_e4m3fn).view(torch.uint8), X.float().to(torch.float8_e4m3fn).view(torch.uint8)
    o32 = torch.empty(Mo, No, dtype=torch.float32, device=dev)
    fl = 2.0 * M * Mo * No
    t16 = timeit(lambda: hb.gemm(dY, X, Mo, No, M, 1, 1, hb.EPI_F32_SPLITK, out=o32))
    import ctypes as C
    ws = torch.empty(max(hb.lib().nbest_wgrad_fp8_ws_bytes(Mo, No, M), 16), dtype=torch.uint8, device=dev)
    def f8():
        hb.check(hb.lib().nbest_wgrad_fp8(hb.ptr(dY8), hb.ptr(X8), hb.ptr(o32), Mo, No, M, Mo, No, No, None, 0, hb.ptr(ws), ws.numel(), hb.stream_ptr()), "wgrad_fp8")
    t8 = timeit(f8)
    print("wgrad %-9s %4dx%4d K=%d   bf16 %7.1f us %6.0f TF/s | fp8 %7.1f us %6.0f TF/s  (x%.2f)" % (nm, Mo, No, M, t16 * 1e3, fl / t16 / 1e9, t8 * 1e3, fl / t8 / 1e9, t16 / t8), flush=True)
