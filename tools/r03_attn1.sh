#!/bin/bash
set -e
out=gpurun_out/r03_attn1; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $out/kernels.log 2>&1 || { tail -30 $out/kernels.log; exit 1; }
echo "[attn1] attention tests ok"
NBEST_LIB=$PWD/scratch_libs/libnbest_r02.so python tools/hbm_bench.py > $out/hbm_r02.log 2>&1
python tools/hbm_bench.py > $out/hbm_new.log 2>&1
grep -h attention $out/hbm_r02.log $out/hbm_new.log
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_new.log 2>&1
grep -h "timed region" $out/bench_new.log
