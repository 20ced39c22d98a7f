#!/bin/bash
# One rocprofv3 evidence set for bench.py's default workload, run ON THE GPU BOX (through gpurun):
#   tools/profile_step.sh r02 [dtype]  -> gpurun_out/r02/{trace,fetch,write}/ + gpurun_out/r02/pmc.csv + kernel_stats.csv
# Three separate runs of the same command: kernel trace (durations), PMC FETCH_SIZE, PMC WRITE_SIZE (the two TCC
# counters do not fit one pass; a --pmc run never carries --kernel-trace's sibling trace domains).
set -e
tag=${1:-r02}
dtype=${2:-bf16}          # tools/profile_step.sh r02_fp8w fp8w -> the fp8 mode's tables
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
ARGS="bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline --dtype $dtype"
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 $ARGS > $out/trace.log 2>&1
echo "[profile_step] kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- python3 $ARGS > $out/fetch.log 2>&1
echo "[profile_step] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE -d $out/write -- python3 $ARGS > $out/write.log 2>&1
echo "[profile_step] WRITE_SIZE pass done"
python3 tools/pmc_table.py $out/trace/*/*.db $out/fetch/*/*.db $out/write/*/*.db 8 > $out/pmc.csv
python3 tools/rocpd_stats.py $out/trace/*/*.db > $out/kernel_stats.csv
rm -rf $out/fetch $out/write       # the databases are large; the CSVs are what is kept
head -25 $out/pmc.csv
