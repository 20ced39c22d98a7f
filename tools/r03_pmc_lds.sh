#!/bin/bash
# LDS bank conflicts / wave stalls of the GEMM kernels: round-2 library vs the register-epilogue build
set -e
out=gpurun_out/r03_pmc_lds; mkdir -p $out
export TMPDIR=/tmp
for lib in r02 new; do
  if [ $lib = r02 ]; then export NBEST_LIB=$PWD/scratch_libs/libnbest_r02.so; else unset NBEST_LIB; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/$lib -- python3 tools/layer_gemms.py --rounds 2 --inner 1 --only "N768  K768" > $out/$lib.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for lib in ("r02", "new"):
    f = glob.glob("gpurun_out/r03_pmc_lds/%s/**/*counter_collection.csv" % lib, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"][:90]
            if "gemm" not in k: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(lib, k)
        for c, v in sorted(d.items()):
            print("    %-24s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
