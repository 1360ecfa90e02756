#!/bin/bash
# rocprofv3 kernel stats of the bf16 graph rollout: once per "VAR=value" setting given as arguments (default: the shipped configuration)
OUT=gpurun_out/${TAG:-lin16_prof}
mkdir -p $OUT
export TMPDIR=/tmp DT=bf16 GRAPH=1
for s in "${@:-X=0}"; do
  export "$s"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$s -- python3 scripts/prof_policy_rollout.py > $OUT/prof_$s.log 2>&1 || exit 1
  F=$(find $OUT/prof_$s -name "*kernel_stats*.csv" | head -1)
  python3 scripts/short_stats.py $F > $OUT/kernel_stats_$s.txt
  rm -rf $OUT/prof_$s
  echo "== $s"; head -8 $OUT/kernel_stats_$s.txt
  unset "${s%%=*}"
done
