#!/bin/bash
# rocprofv3 kernel trace of scripts/fair_step_probe.py -> the kernel-by-kernel timeline of one FusedFair minibatch step
# usage: bash scripts/prof_fair_step.sh <out file>
OUT=$1
export TMPDIR=/tmp
D=$(mktemp -d /tmp/prof_fair.XXXX)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 scripts/fair_step_probe.py fused_update=1 > $D/log.txt 2>&1 || { tail -5 $D/log.txt; exit 1; }
python3 scripts/update_timeline.py $(find $D -name "*kernel_trace*.csv" | head -1) -1 k_shard_apply > $OUT
rm -rf $D
cat $OUT
