#!/bin/bash
# rocprofv3 kernel trace of scripts/step_ab.py for ONE variant -> the kernel-by-kernel timeline of one minibatch step
# usage: bash scripts/prof_step.sh <out file> <variant>     e.g. bash scripts/prof_step.sh gpurun_out/r04c/tl_own1.txt own_gemm=1
#        LAST=k_shard_apply bash scripts/prof_step.sh <out file> force_collectives=1,grad_allreduce=flat rccl=1   (the multi-rank program at world 1:
#        real RCCL nodes in the graph; its step ends with k_shard_apply)
OUT=$1; shift
export TMPDIR=/tmp
D=$(mktemp -d /tmp/prof_step.XXXX)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 scripts/step_ab.py "$@" rounds=1 steps=64 > $D/log.txt 2>&1 || { tail -5 $D/log.txt; exit 1; }
python3 scripts/update_timeline.py $(find $D -name "*kernel_trace*.csv" | head -1) -1 ${LAST:-k_adam_apply} > $OUT
rm -rf $D
cat $OUT
