#!/bin/bash
set -e
D0=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
D1=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag1024.so
out=gpurun_out/r03_aseg; mkdir -p $out
NBEST_LIB=$D0 python tools/layer_gemms.py --tag base > $out/base.log 2>&1
NBEST_LIB=$D1 python tools/layer_gemms.py --tag aseg > $out/aseg.log 2>&1
NBEST_LIB=$D0 python tools/layer_gemms.py --tag base2 > $out/base2.log 2>&1
for t in base aseg base2; do echo "== $t"; grep -E "fwd|dgrd" $out/$t.log; done
