#!/bin/bash
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
echo "== base"; NBEST_LIB=$D python tools/gemm_ksweep.py
echo "== sym"; NBEST_LIB=$D NBEST_SYM=1 python tools/gemm_ksweep.py
