#!/bin/bash
# tile columns per L2 group (NBEST_GN, diag build) with packed weight operands: one layer's forward / dgrad GEMMs
set -e
D=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so
out=gpurun_out/ab_gn; mkdir -p $out
for g in rule 1 2 3 4 0 rule; do
  if [ $g = rule ]; then unset NBEST_GN; else export NBEST_GN=$g; fi
  NBEST_LIB=$D python tools/layer_gemms.py --tag gn$g > $out/gn$g.log 2>&1
  echo "== gn=$g: $(grep -E 'fwd|dgrd' $out/gn$g.log | awk '{for(i=1;i<=NF;i++) if($i=="median"){s+=$(i+1); printf "%s ", $(i+1)}} END{print " sum", s}')"
done
