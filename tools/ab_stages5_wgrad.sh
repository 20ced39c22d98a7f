#!/bin/bash
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/ab_stages5w; mkdir -p $out
NBEST_LIB=$D NBEST_STAGES=5 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad_pair or bert_shapes or layouts" > $out/ktest.log 2>&1 || { tail -20 $out/ktest.log; exit 1; }
tail -1 $out/ktest.log
for t in s4 s5 s4b s5b; do
  case $t in s5*) export NBEST_STAGES=5;; *) unset NBEST_STAGES;; esac
  NBEST_LIB=$D python tools/layer_gemms.py --only wgrd --tag $t > $out/$t.log 2>&1
  echo "== $t: $(grep -E 'wgrd' $out/$t.log | awk '{for(i=1;i<=NF;i++) if($i=="median"){printf "%s ", $(i+1)}}')"
done
