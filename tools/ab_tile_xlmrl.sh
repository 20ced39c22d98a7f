#!/bin/bash
# xlm-roberta-large shape (M = 16 384 token rows): default tile plan vs forced tiles (diag build)
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/ab_tile_xlmrl; mkdir -p $out
for t in default 256x256 256x128 256x192; do
  if [ $t = default ]; then NBEST_LIB=$D python bench.py --model xlm-roberta-large --batch 64 --seq_len 256 --no_cpu_baseline --no_roofline --steps 10 --warmup 3 > $out/$t.log 2>&1
  else NBEST_LIB=$D NBEST_TILE=$t python bench.py --model xlm-roberta-large --batch 64 --seq_len 256 --no_cpu_baseline --no_roofline --steps 10 --warmup 3 > $out/$t.log 2>&1; fi
  echo "== $t: $(grep 'timed region' $out/$t.log)"
done
