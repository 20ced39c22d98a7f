#!/bin/bash
# 4-stage vs 5-stage LDS ring of the k-contiguous ping-pong kernels (diag build, NBEST_STAGES=5), one layer's GEMMs, interleaved
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/ab_stages5; mkdir -p $out
NBEST_LIB=$D NBEST_STAGES=5 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $out/ktest.log 2>&1 || { tail -20 $out/ktest.log; exit 1; }
tail -1 $out/ktest.log
for t in s4 s5 s4b s5b; do
  case $t in s5*) export NBEST_STAGES=5;; *) unset NBEST_STAGES;; esac
  NBEST_LIB=$D python tools/layer_gemms.py --tag $t > $out/$t.log 2>&1
  echo "== $t: $(grep -E 'fwd|dgrd' $out/$t.log | awk '{s+=$(NF-7)} END{print s}') us  $(grep -E 'fwd|dgrd' $out/$t.log | awk '{printf "%s ", $(NF-7)}')"
done
