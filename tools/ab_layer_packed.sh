#!/bin/bash
set -e
out=gpurun_out/r03_packiso; mkdir -p $out
python tools/layer_gemms.py --no_pack --tag nopack > $out/nopack.log 2>&1
python tools/layer_gemms.py --tag pack > $out/pack.log 2>&1
python tools/layer_gemms.py --no_pack --tag nopack2 > $out/nopack2.log 2>&1
for t in nopack pack nopack2; do echo "== $t"; grep -E "fwd|dgrd" $out/$t.log | awk '{s+=$(NF-7)} END{print "sum of medians", s}'; grep -E "fwd|dgrd" $out/$t.log | cut -c1-75; done
