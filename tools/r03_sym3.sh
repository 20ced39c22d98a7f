#!/bin/bash
set -e
export TMPDIR=/tmp
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/r03_sym3; mkdir -p $out
NBEST_LIB=$D NBEST_SYM=1 rocprofv3 --kernel-trace --stats -d $out/trace -- python3 tools/gemm_ksweep.py > $out/sweep.log 2>&1
python3 tools/rocpd_stats.py $out/trace/*/*.db > $out/stats.csv
python3 tools/kernel_sequence.py $out/trace/*/*.db 40 > $out/seq.txt
rm -rf $out/trace
cat $out/stats.csv | cut -c1-200 | head; tail -24 $out/seq.txt
