#!/bin/bash
# rocprofv3 --pmc passes (separate, kernel-trace only) on the default fused rollout kernel: instruction mix and wave activity
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mix
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 scripts/pmc_run.py > $OUT/p$i.log 2>&1
done
python3 scripts/pmc_show.py $OUT
