#!/bin/bash
set -e
out=gpurun_out/r03_seq; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_roofline > $out/trace.log 2>&1
python3 tools/rocpd_stats.py $out/trace/*/*.db > $out/kernel_stats.csv
python3 tools/kernel_sequence.py $out/trace/*/*.db 420 > $out/sequence.txt
rm -rf $out/trace
grep -i "heads\|copyBuffer\|fill" $out/kernel_stats.csv | cut -c1-160
