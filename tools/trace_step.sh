#!/bin/bash
# kernel trace of the default bench workload only (no PMC passes): tools/trace_step.sh <tag> [dtype]
set -e
tag=${1:-trace}; dtype=${2:-bf16}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline --dtype $dtype > $out/trace.log 2>&1
python3 tools/rocpd_stats.py $out/trace/*/*.db > $out/kernel_stats.csv
rm -rf $out/trace
head -45 $out/kernel_stats.csv
