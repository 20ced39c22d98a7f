#!/bin/bash
# Where the GEMM kernels' wave time goes, IN the training step and in ISOLATION (run on the GPU box through gpurun):
#   tools/pmc_sq.sh r04   -> gpurun_out/r04_sq/{step,alone}.csv  (tools/pmc_sq_table.py)
# One rocprofv3 pass per arm: 8 SQ counters + GRBM_GUI_ACTIVE (the clock: MI355X_MICROARCH.md, DVFS give-back) with --kernel-trace for
# the durations of the SAME dispatches (a profiled pass runs at its own clock: never mix arms).
set -e
tag=${1:-r04}
out=gpurun_out/${tag}_sq
mkdir -p $out
export TMPDIR=/tmp
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $out/step -- python3 bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline > $out/step.log 2>&1
echo "[pmc_sq] in-step pass done"
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $out/alone -- python3 tools/layer_gemms.py --rounds 6 --inner 2 > $out/alone.log 2>&1
echo "[pmc_sq] isolated pass done"
python3 tools/pmc_sq_table.py $out/step > $out/step.csv
python3 tools/pmc_sq_table.py $out/alone > $out/alone.csv
rm -rf $out/step $out/alone
head -12 $out/step.csv; head -12 $out/alone.csv
