#!/bin/bash
# LDS bank conflicts / wave stalls of the k-contiguous GEMM kernels: diag build (A staged per stage) vs the current library (A staged in pairs)
set -e
out=gpurun_out/r03_pmc_lds2; mkdir -p $out
export TMPDIR=/tmp
for lib in old new; do
  if [ $lib = old ]; then export NBEST_LIB=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so; else unset NBEST_LIB; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/$lib -- python3 tools/layer_gemms.py --rounds 2 --inner 1 --only "fwd  qkv" > $out/$lib.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for lib in ("old", "new"):
    f = glob.glob("gpurun_out/r03_pmc_lds2/%s/**/*counter_collection.csv" % lib, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"][:90]
            if "gemm2" not in k: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(lib, k)
        for c, v in sorted(d.items()):
            print("    %-24s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
