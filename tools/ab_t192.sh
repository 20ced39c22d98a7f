#!/bin/bash
# 256x192 tiles for the N = 768 GEMMs from ONE full round of tiles (NBEST_T192MIN=256, diag build) instead of two: M = 16 384 shapes
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
for args in "--batch 64 --seq_len 256 --n_best 10 --add_l2_loss" "--batch 128"; do
for t in 512 256 512 256; do
  echo "== $args  t192_min=$t: $(NBEST_LIB=$D NBEST_T192MIN=$t python bench.py $args --no_cpu_baseline --no_roofline --steps 10 --warmup 3 2>&1 | grep 'timed region')"
done; done
