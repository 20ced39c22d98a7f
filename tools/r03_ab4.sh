#!/bin/bash
set -e
out=gpurun_out/r03_ab4; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $out/kernels.log 2>&1 || { tail -30 $out/kernels.log; exit 1; }
echo "[ab4] kernel tests ok"
NBEST_LIB=$PWD/scratch_libs/libnbest_r02.so python tools/layer_gemms.py --tag r02 > $out/gemms_r02.log 2>&1
python tools/layer_gemms.py --tag new > $out/gemms_new.log 2>&1
echo "[ab4] gemm A/B done"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_new.log 2>&1
NBEST_LIB=$PWD/scratch_libs/libnbest_r02.so python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_r02.log 2>&1
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_new2.log 2>&1
grep -h "timed region" $out/bench_*.log
