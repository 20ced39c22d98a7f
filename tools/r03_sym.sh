#!/bin/bash
set -e
out=gpurun_out/r03_sym; mkdir -p $out
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
NBEST_LIB=$D NBEST_SYM=1 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $out/ktest.log 2>&1 || { tail -20 $out/ktest.log; exit 1; }
tail -2 $out/ktest.log
NBEST_LIB=$D python tools/layer_gemms.py --tag base > $out/base.log 2>&1
NBEST_LIB=$D NBEST_SYM=1 python tools/layer_gemms.py --tag sym > $out/sym.log 2>&1
NBEST_LIB=$D python tools/layer_gemms.py --tag base2 > $out/base2.log 2>&1
for t in base sym base2; do echo "== $t"; grep -E "fwd  qkv|fwd  ffn-up|dgrd ffn-down" $out/$t.log; done
