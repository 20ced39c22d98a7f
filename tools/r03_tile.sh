#!/bin/bash
set -e
out=gpurun_out/r03_tile; mkdir -p $out
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
for t in default 256x128 256x192; do
  if [ $t = default ]; then NBEST_LIB=$D python tools/layer_gemms.py --tag $t > $out/$t.log 2>&1
  else NBEST_LIB=$D NBEST_TILE=$t python tools/layer_gemms.py --tag $t > $out/$t.log 2>&1; fi
  echo "== $t"; grep -E "fwd|dgrd" $out/$t.log
done
