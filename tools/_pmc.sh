mkdir -p gpurun_out/r9
export TMPDIR=/tmp
run() { # name, env, args
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
    g=$(echo $grp | cut -c1-12 | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/r9/$1_$g -- python tools/gemm_one.py $2 4 > /dev/null 2>&1
  done
}
NBEST_TILE=256x256 run ppTN wgrad
run v1TN wgrad
run ppNT ffn_up
ls gpurun_out/r9 | head -20
