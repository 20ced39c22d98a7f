#!/bin/bash
set -e
out=gpurun_out/ab_step_packed_fp8; mkdir -p $out
for i in 1 2; do
  python bench.py --dtype fp8w --no_cpu_baseline --no_roofline --no_packed_weights > $out/nopack$i.log 2>&1
  python bench.py --dtype fp8w --no_cpu_baseline --no_roofline > $out/pack$i.log 2>&1
done
grep -h "timed region" $out/nopack1.log $out/pack1.log $out/nopack2.log $out/pack2.log
