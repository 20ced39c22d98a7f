#!/bin/bash
# the round-1 persistent ping-pong kernel (LDS-restaged epilogue, diag build, NBEST_PERSISTENT=1) against the shipped per-tile kernel
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/ab_persistent; mkdir -p $out
for t in tile pers tile2 pers2; do
  case $t in pers*) export NBEST_PERSISTENT=1;; *) unset NBEST_PERSISTENT;; esac
  NBEST_LIB=$D python tools/layer_gemms.py --only "fwd" --tag $t > $out/$t.log 2>&1
  echo "== $t: $(grep -E 'fwd' $out/$t.log | awk '{for(i=1;i<=NF;i++) if($i=="median"){printf "%s ", $(i+1)}}')"
done
