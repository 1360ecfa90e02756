#!/bin/bash
# rocprofv3 kernel trace of the policy-in-the-loop ROLLOUT alone (bf16 inference, hipGraph-replayed) + phase timings
TAG=${1:-r02_rollout_policy}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/bench_policy.py 2>/dev/null | tail -1 | tee $OUT/policy_path_phases.json
DT=bf16 GRAPH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 scripts/prof_policy_rollout.py > $OUT/prof.log 2>&1
F=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1)
python3 scripts/short_stats.py $F | tee $OUT/kernel_stats_rollout_bf16_graph.txt | head -24
rm -rf $OUT/prof
