#!/bin/bash
# LayerNorm backward with 8 instead of 4 waves per block (library built with -DNBEST_LN_WAVES=8), in the step (kernel trace), alternating
set -e
export TMPDIR=/tmp
for v in w4 w8 w4b w8b; do
  case $v in w8*) export NBEST_LIB=$PWD/gpurun_scratch/libnbest_ln8.so;; *) unset NBEST_LIB;; esac
  out=gpurun_out/ab_ln_waves/$v; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline > $out/trace.log 2>&1
  python3 tools/rocpd_stats.py $out/trace/*/*.db > $out/kernel_stats.csv; rm -rf $out/trace
  echo "== $v: $(grep 'timed region' $out/trace.log)  ln_bwd avg ns: $(grep ln_bwd_fast $out/kernel_stats.csv | awk -F, '{print $(NF-3)}')"
done
