#!/bin/bash
set -e
out=gpurun_out/r03_attn2; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $out/kernels.log 2>&1 || { tail -40 $out/kernels.log; exit 1; }
echo "[attn2] attention tests ok"
python tools/hbm_bench.py > $out/hbm_new.log 2>&1
grep -h attention $out/hbm_new.log
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_new.log 2>&1
grep -h "timed region" $out/bench_new.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "reference_outputs or edge or long or capturable or additivity or trajectory" > $out/model.log 2>&1 || { tail -40 $out/model.log; exit 1; }
tail -2 $out/model.log
