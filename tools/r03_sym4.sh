#!/bin/bash
set -e
for d in 64 128 256 512; do
  D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag$d.so
  echo "== sym DIAG=$d"; NBEST_LIB=$D NBEST_SYM=1 python tools/gemm_ksweep.py | grep -E "K   128|K   768|K  3072"
done
