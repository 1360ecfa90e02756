#!/bin/bash
# rocprofv3 --pmc passes (separate, kernel-trace only) over the PPO minibatch step (scripts/step_ab.py, one variant): per kernel,
# MFMA utilisation / fp32 MFMA flops, LDS activity and bank conflicts, HBM-side bytes.
# usage: bash scripts/pmc_step.sh <out file> [variant]      e.g. bash scripts/pmc_step.sh gpurun_out/r04u/step_pmc.txt own_gemm=1
#        PMC_FAIR=1 bash scripts/pmc_step.sh <out file> [variant]   the FAIR network's step (scripts/fair_step_probe.py, e.g. fair_chain=1)
OUTF=$1; shift
VAR=${1:-own_gemm=1}
if [ -n "$PMC_FAIR" ]; then PROG="scripts/fair_step_probe.py $VAR"; else PROG="scripts/step_ab.py $VAR rounds=1 steps=16"; fi
export TMPDIR=/tmp
D=$(mktemp -d /tmp/pmc_step.XXXX)
i=0
for set in "MfmaUtil" "MfmaFlopsF32 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/p$i -- python3 $PROG > $D/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $D/p$i.log; }
done
python3 scripts/pmc_step_show.py $D > $OUTF
rm -rf $D
cat $OUTF
