#!/bin/bash
# round-3 experiment 1: direct (register) GEMM epilogues vs the round-2 library, tile variants for the N = 768 shapes
set -e
out=gpurun_out/r03_ab1; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $out/kernels.log 2>&1 || { tail -30 $out/kernels.log; exit 1; }
echo "[ab1] kernel tests ok"
NBEST_LIB=$PWD/scratch_libs/libnbest_r02.so python tools/layer_gemms.py --tag r02 > $out/gemms_r02.log 2>&1
python tools/layer_gemms.py --tag direct > $out/gemms_new.log 2>&1
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
for t in 256x256 256x128 128x256 128x128; do
  NBEST_LIB=$D NBEST_TILE=$t python tools/layer_gemms.py --tag tile$t --only N768 > $out/gemms_$t.log 2>&1
done
NBEST_LIB=$D NBEST_GEMM=v1 python tools/layer_gemms.py --tag v1 --only "fwd" > $out/gemms_v1fwd.log 2>&1
echo "[ab1] gemm A/B done"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_new.log 2>&1
NBEST_LIB=$PWD/scratch_libs/libnbest_r02.so python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_r02.log 2>&1
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_new2.log 2>&1
grep -h "timed region" $out/bench_*.log
