#!/bin/bash
set -e
D0=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/r03_apair; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $out/ktest.log 2>&1 || { tail -30 $out/ktest.log; exit 1; }
tail -2 $out/ktest.log
NBEST_LIB=$D0 python tools/layer_gemms.py --tag old > $out/old.log 2>&1
python tools/layer_gemms.py --tag new > $out/new.log 2>&1
NBEST_LIB=$D0 python tools/layer_gemms.py --tag old2 > $out/old2.log 2>&1
python tools/layer_gemms.py --tag new2 > $out/new2.log 2>&1
for t in old new old2 new2; do echo "== $t"; grep -E "fwd|dgrd" $out/$t.log | cut -c1-75; done
