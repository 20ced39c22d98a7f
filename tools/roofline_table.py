#!/usr/bin/env python3
"""Per-kernel roofline table (markdown) of one bench.py step from profiles/rNN_pmc.csv.

    python tools/roofline_table.py profiles/r02_pmc.csv

Shapes are those of bench.py's default workload (bert-base, 256 x 128 tokens): flops / algorithmic bytes per launch are
computed here from the shapes, durations and HBM-side traffic come from the CSV (rocprofv3, tools/profile_step.sh).
Peaks: 2 500 TFLOP/s dense bf16 MFMA, 8 TB/s HBM3E (MI355X_MICROARCH.md)."""
import csv
import sys

M, H, F, B, S, HEADS, L = 32768, 768, 3072, 256, 128, 12, 12
PEAK_TF, PEAK_TB = 2500.0, 8.0
PEAK_TF8 = 5000.0          # dense fp8 MFMA: the rows of the fp8w mode's GEMMs (marked "fp8") are priced against it


def gf(m, n, k):
    return 2.0 * m * n * k / 1e9


# (substring of the kernel name, grid size or None) -> (label, GFLOP per launch or None, algorithmic MB per launch)
# for kernels that serve two shapes through one (name, grid) the figures are the per-launch average
KERNELS = [
    # ---- fp8w mode (profiles/rNN_fp8w_pmc.csv): e4m3 operands, 1 byte each ----
    ("gemm8tt_kernel", 129024, "fp8 weight gradient FFN-up / FFN-down (3072x768 | 768x3072, K = 32 768)", gf(F, H, M), (M * (F + H) + F * H * 4) / 1e6),
    ("gemm8tt_kernel", 124416, "fp8 weight gradient QKV (2304x768) | attention-out (768x768)", (gf(3 * H, H, M) + gf(H, H, M)) / 2, (M * (4 * H + 2 * H) / 2 + (3 * H * H + H * H) / 2 * 4) / 1e6),
    ("gemm8_kernel<5, 4>", None, "fp8 dgrad + residual: QKV (K 2304) | FFN-up (K 3072), N 768", (gf(M, H, 3 * H) + gf(M, H, F)) / 2, (M * (3 * H + F) / 2 + 2 * M * H * 2 + (3 * H + F) / 2 * H) / 1e6),
    ("gemm8_kernel<2, 4>", None, "fp8 forward FFN-up + bias + GELU (writes GELU' 8-bit and e4m3 GELU)", gf(M, F, H), (M * H + 2 * M * F + F * H) / 1e6),
    ("gemm8_kernel<4, 4>", None, "fp8 dgrad FFN-down x GELU' (+ fused db1; e4m3 output only)", gf(M, F, H), (M * H + 2 * M * F + F * H) / 1e6),
    ("gemm8_kernel<3, 4>", None, "fp8 forward FFN-down + bias + dropout + residual", gf(M, H, F), (M * F + 2 * M * H * 2 + F * H) / 1e6),
    ("gemm8_kernel<3, 2>", None, "fp8 forward attention-out + bias + dropout + residual (256x128 tiles)", gf(M, H, H), (M * H + 2 * M * H * 2 + H * H) / 1e6),
    ("gemm8_kernel<1, 4>", None, "fp8 forward QKV + bias", gf(M, 3 * H, H), (M * H + M * 3 * H * 2 + 3 * H * H) / 1e6),
    ("gemm8_kernel<0, 2>", None, "fp8 dgrad attention-out (256x128 tiles)", gf(M, H, H), (M * H + M * H * 2 + H * H) / 1e6),
    ("splitk_reduce8_kernel", None, "split-K reduce of an fp8 weight gradient", None, None),
    ("quant_w8t_kernel", None, "e4m3 transposed weight copy (from the fp32 master)", None, 85e6 * 5 / 1e6),
    ("quant_w8_kernel", None, "e4m3 weight copy (from the fp32 master)", None, 85e6 * 5 / 1e6),
    ("absmax_kernel", None, "per-matrix max |w| (weight scales)", None, 85e6 * 4 / 1e6),
    # ---- bf16 ----
    ("gemm2_kernel<256, 256, 2, 4, 4, true, true, 6>", 129024, "weight gradients, K = 32 768: FFN-down 768x3072 | FFN-up 3072x768 | QKV 2304x768 + attention-out 768x768 in one launch (round 3; 36 tiles x 7 K-splits each)", gf(F, H, M), (M * (F + H) * 2 + F * H * 4) / 1e6),
    ("gemm2_kernel<256, 256, 2, 4, 4, true, true, 6>", 124416, "weight gradient QKV (2304x768)", gf(3 * H, H, M), (M * 4 * H * 2 + 3 * H * H * 4) / 1e6),
    ("gemm_bf16_kernel<true, true, 6>", None, "weight gradient attention-out (768x768)", gf(H, H, M), (M * 2 * H * 2 + H * H * 4) / 1e6),
    # round 3: register-epilogue kernels (256 x 256 tiles as 4 x 2 waves; 256 x 192 tiles for the N = 768 shapes)
    # round 4: 128 x 384 tiles (2 x 4 waves of 64 x 96) took the N = 768 shapes over from 256 x 192
    ("gemm2_kernel<128, 384, 2, 4, 5, false, false, 5>", None, "dgrad + residual: QKV (K 2304) | FFN-up (K 3072), N 768 (128x384 tiles, 5-stage ring)", (gf(M, H, 3 * H) + gf(M, H, F)) / 2, (M * (3 * H + F) / 2 * 2 + 2 * M * H * 2 + (3 * H + F) / 2 * H * 2) / 1e6),
    ("gemm2_kernel<128, 384, 2, 4, 5, false, false, 3>", None, "forward FFN-down + bias + dropout + residual (K 3072; 128x384 tiles, 5-stage ring)", gf(M, H, F), (M * F * 2 + 2 * M * H * 2 + F * H * 2) / 1e6),
    ("gemm2_kernel<128, 384, 2, 4, 4, false, false, 3>", None, "forward attention-out + bias + dropout + residual (K 768; 128x384 tiles)", gf(M, H, H), (M * H * 2 + 2 * M * H * 2 + H * H * 2) / 1e6),
    ("gemm2_kernel<128, 384, 2, 4, 4, false, false, 0>", None, "dgrad attention-out (128x384 tiles)", gf(M, H, H), (2 * M * H * 2 + H * H * 2) / 1e6),
    ("gemm2_kernel<256, 192, 4, 2, 5, false, false, 5>", None, "dgrad + residual: QKV (K 2304) | FFN-up (K 3072), N 768 (256x192 tiles, 5-stage ring)", (gf(M, H, 3 * H) + gf(M, H, F)) / 2, (M * (3 * H + F) / 2 * 2 + 2 * M * H * 2 + (3 * H + F) / 2 * H * 2) / 1e6),
    ("gemm2_kernel<256, 192, 4, 2, 5, false, false, 3>", None, "forward FFN-down + bias + dropout + residual (K 3072; 256x192 tiles, 5-stage ring)", gf(M, H, F), (M * F * 2 + 2 * M * H * 2 + F * H * 2) / 1e6),
    ("gemm2_kernel<256, 192, 4, 2, 4, false, false, 5>", None, "dgrad + residual: QKV (K 2304) | FFN-up (K 3072), N 768 (256x192 tiles)", (gf(M, H, 3 * H) + gf(M, H, F)) / 2, (M * (3 * H + F) / 2 * 2 + 2 * M * H * 2 + (3 * H + F) / 2 * H * 2) / 1e6),
    ("gemm2_kernel<256, 192, 4, 2, 4, false, false, 3>", None, "forward bias+dropout+residual: attention-out (K 768) | FFN-down (K 3072) (256x192 tiles)", (gf(M, H, H) + gf(M, H, F)) / 2, (M * (H + F) / 2 * 2 + 2 * M * H * 2) / 1e6),
    ("gemm2_kernel<256, 192, 4, 2, 4, false, false, 0>", None, "dgrad attention-out (256x192 tiles)", gf(M, H, H), (2 * M * H * 2 + H * H * 2) / 1e6),
    ("gemm2_kernel<256, 256, 4, 2, 4, false, false, 2>", None, "forward FFN-up + bias + GELU (writes GELU and 8-bit GELU')", gf(M, F, H), (M * H * 2 + M * F * 3 + F * H * 2) / 1e6),
    ("gemm2_kernel<256, 256, 4, 2, 4, false, false, 4>", None, "dgrad FFN-down x GELU' (+ fused db1)", gf(M, F, H), (M * H * 2 + M * F * 3 + F * H * 2) / 1e6),
    ("gemm2_kernel<256, 256, 4, 2, 4, false, false, 1>", None, "forward QKV + bias", gf(M, 3 * H, H), (M * H * 2 + M * 3 * H * 2 + 3 * H * H * 2) / 1e6),
    ("gemm_bf16_kernel<false, false, 5>", None, "dgrad + residual: QKV (K 2304) | FFN-up (K 3072), N 768", (gf(M, H, 3 * H) + gf(M, H, F)) / 2, (M * (3 * H + F) / 2 * 2 + 2 * M * H * 2 + (3 * H + F) / 2 * H * 2) / 1e6),
    ("gemm_bf16_kernel<false, false, 3>", None, "forward bias+dropout+residual: attention-out (K 768) | FFN-down (K 3072)", (gf(M, H, H) + gf(M, H, F)) / 2, (M * (H + F) / 2 * 2 + 2 * M * H * 2) / 1e6),
    ("gemm2_kernel<256, 256, 2, 4, 4, false, false, 2>", None, "forward FFN-up + bias + GELU (writes GELU and GELU')", gf(M, F, H), (M * H * 2 + 2 * M * F * 2 + F * H * 2) / 1e6),
    ("gemm2_kernel<256, 256, 2, 4, 4, false, false, 4>", None, "dgrad FFN-down x GELU' (+ fused db1)", gf(M, F, H), (M * H * 2 + 2 * M * F * 2 + F * H * 2) / 1e6),
    ("gemm2_kernel<256, 256, 2, 4, 4, false, false, 1>", None, "forward QKV + bias", gf(M, 3 * H, H), (M * H * 2 + M * 3 * H * 2 + 3 * H * H * 2) / 1e6),
    ("gemm_bf16_kernel<false, false, 0>", None, "dgrad attention-out", gf(M, H, H), (2 * M * H * 2 + H * H * 2) / 1e6),
    ("attn_bwd3_bf16_kernel", None, "attention backward, 16-key waves (recompute P; dQ dK dV + bias gradient)", 5 * 4.0 * S * S * 64 * B * HEADS / 2 / 1e9 * 1.0, (M * 3 * H * 2 * 2 + 2 * M * H * 2) / 1e6),
    ("attn_bwd2_bf16_kernel", None, "attention backward (recompute P; dQ dK dV + bias gradient)", 5 * 4.0 * S * S * 64 * B * HEADS / 2 / 1e9 * 1.0, (M * 3 * H * 2 * 2 + 2 * M * H * 2) / 1e6),
    ("attn_fwd_bf16_kernel", None, "attention forward", 2 * 2.0 * S * S * 64 * B * HEADS / 1e9, (M * 3 * H * 2 + M * H * 2) / 1e6),
    ("ln_bwd_fast_kernel", None, "LayerNorm backward (+ dropout mask, dgamma dbeta dbias partials)", None, (4 * M * H * 2) / 1e6),
    ("ln_fwd_fast_kernel", None, "LayerNorm forward", None, (2 * M * H * 2) / 1e6),
    ("bertadam_kernel", 1374208, "BertAdam, encoder layers + heads (85 M parameters)", None, 85.6e6 * 30 / 1e6),
    ("bertadam_kernel", 373248, "BertAdam, embedding tables (23.8 M parameters)", None, 23.8e6 * 30 / 1e6),
    ("sumsq_kernel", 1374208, "per-tensor gradient norms (clip)", None, 85.6e6 * 4 / 1e6),
    ("splitk_reduce2_kernel", None, "split-K reduce of a weight gradient", None, None),
    ("splitk_reduce_kernel", None, "split-K reduce (768x768)", None, None),
    ("pack_b_kernel", None, "weight matrices packed for the k-contiguous GEMMs (forward copy, dgrad copy)", None, 85e6 * 4 / 1e6),
    ("transpose_multi_kernel", None, "k-contiguous bf16 weight copy for the dgrads", None, 85e6 * 4 / 1e6),
    ("embed_bwd_tables_kernel", None, "embedding backward: position / type tables, LN parameters (folded into embed_bwd_kernel in round 3)", None, None),
    ("embed_bwd_kernel", None, "embedding backward: LN-bwd + word-table atomics", None, None),
    ("rowred_finalize_multi_kernel", None, "partial-row sums of a backward layer -> bias / LayerNorm gradients (one launch per layer)", None, None),
    ("rowred_finalize_kernel", None, "partial-row sums -> bias / LayerNorm gradients", None, None),
]


def main(path):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["ms_per_step"]) for r in rows)
    print("| ms/step | launches | avg µs | kernel | algorithmic | achieved | of peak | HBM-side traffic (PMC) |")
    print("|---|---|---|---|---|---|---|---|")
    shown = 0.0
    for r in rows:
        name, grid = r["kernel"], int(r["grid_size"])
        if name.startswith("gemm2_kernel<"):          # the trailing SYM template flag (experiment builds only) is not part of the patterns
            name = name.replace(", false>", ">")
        hit = None
        for sub, g, label, flops, mb in KERNELS:
            if sub in name and (g is None or g == grid):
                hit = (label, flops, mb)
                break
        if hit is None:
            continue
        label, flops, mb = hit
        if "256, 192, 4, 2, 4, false, false, 3>" in name and float(r["launches_per_step"]) < 18:   # since the 5-stage ring took FFN-down: attention-out only
            label, flops, mb = "forward attention-out + bias + dropout + residual (K 768; 256x192 tiles)", gf(M, H, H), (M * H * 2 + 2 * M * H * 2 + H * H * 2) / 1e6
        label = label.replace(" | ", " / ")
        us, ms, n = float(r["avg_us"]), float(r["ms_per_step"]), float(r["launches_per_step"])
        traffic = float(r["traffic_bytes_per_launch"]) / 1e6
        if flops:
            ach = flops / us * 1e-3        # GF / µs = PF... -> TFLOP/s = GF / (µs * 1e-6) / 1e3
            ach = flops / (us * 1e-6) / 1e3
            pk = PEAK_TF8 if label.startswith("fp8 ") else PEAK_TF
            alg, a, f = "%.1f GFLOP" % flops, "%.0f TFLOP/s" % ach, "%.1f %% (%sMFMA)" % (100 * ach / pk, "fp8 " if pk == PEAK_TF8 else "")
        elif mb:
            tb = mb / us / 1e6 * 1e6 / 1e6  # MB / µs = TB/s
            tb = mb / us
            alg, a, f = "%.0f MB" % mb, "%.2f TB/s" % tb, "%.0f %% (HBM)" % (100 * tb / PEAK_TB)
        else:
            tb = traffic / us
            alg, a, f = "-", "%.2f TB/s (measured bytes)" % tb, "%.0f %% (HBM)" % (100 * tb / PEAK_TB)
        shown += ms
        import re
        mm = re.search(r"\d+([a-z_0-9]+_kernel)", name) if name.startswith("_Z") else None
        shortname = mm.group(1) if mm else name.split("<")[0]
        print("| %.3f | %.0f | %.1f | `%s` - %s | %s | %s | %s | %.0f MB |" % (ms, n, us, shortname + ("<…>" if "<" in name or mm else ""), label, alg, a, f, traffic))
    print()
    print("Listed: %.2f of %.2f ms of kernel time per step." % (shown, total))


if __name__ == "__main__":
    main(sys.argv[1])
