#!/bin/bash
# persistent ping-pong kernel with the register epilogue (diag build, NBEST_PERSISTENT=2) against the shipped per-tile kernel
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/ab_persistent2; mkdir -p $out
NBEST_LIB=$D NBEST_PERSISTENT=2 timeout -k 10 150 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "register_epilogues_full_size or packed_weights or bert_shapes" > $out/ktest.log 2>&1 || { tail -30 $out/ktest.log; exit 1; }
tail -1 $out/ktest.log
for t in tile pers tile2 pers2; do
  case $t in pers*) export NBEST_PERSISTENT=2;; *) unset NBEST_PERSISTENT;; esac
  NBEST_LIB=$D timeout -k 10 120 python tools/layer_gemms.py --only "fwd" --tag $t > $out/$t.log 2>&1
  echo "== $t: $(grep -E 'fwd' $out/$t.log | awk '{for(i=1;i<=NF;i++) if($i=="median"){printf "%s ", $(i+1)}}')"
done
