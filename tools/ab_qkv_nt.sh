#!/bin/bash
# QKV projection output with / without the streaming-store hint: its own time and the attention forward's, in the step (kernel trace)
set -e
export TMPDIR=/tmp
for v in nt plain nt2 plain2; do
  case $v in nt*) L=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag.so;; *) L=$PWD/n-best-asr-transformer_amd/csrc/diag/libnbest_diag4096.so;; esac
  out=gpurun_out/ab_qkv_nt/$v; mkdir -p $out
  NBEST_LIB=$L rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline > $out/trace.log 2>&1
  python3 tools/rocpd_stats.py $out/trace/*/*.db > $out/kernel_stats.csv; rm -rf $out/trace
  echo "== $v: $(grep 'timed region' $out/trace.log)"
  grep -E "false, false, 1|attn_fwd|attn_bwd2" $out/kernel_stats.csv | awk -F, '{print "   ", substr($1,1,70), $(NF-3)}'
done
