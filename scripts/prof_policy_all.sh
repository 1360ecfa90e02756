#!/bin/bash
# Policy-path evidence of a round in one go: bench.py --config ppo (fp32 + bf16), rocprofv3 kernel stats of it, the timeline of one
# minibatch step, rocprofv3 kernel stats of the bf16 graph rollout, head kernels alone, the layer kernel against the library GEMM.
# usage: bash scripts/prof_policy_all.sh <tag>
TAG=${1:-policy}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for dt in bf16 fp32; do
  BRL_INFER_DTYPE=$dt timeout -k 10 600 python bench.py --config ppo --steps 3 2>/dev/null | tail -1 > $OUT/bench_ppo_$dt.json || exit 1
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --config ppo --steps 2 > $OUT/prof.log 2>&1 || exit 1
python3 scripts/short_stats.py $(find $OUT/prof -name "*kernel_stats*.csv" | head -1) > $OUT/kernel_stats_ppo.txt
python3 scripts/update_timeline.py $(find $OUT/prof -name "*kernel_trace*.csv" | head -1) > $OUT/update_timeline.txt
rm -rf $OUT/prof
DT=bf16 GRAPH=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 scripts/prof_policy_rollout.py > $OUT/prof_rollout.log 2>&1 || exit 1
python3 scripts/short_stats.py $(find $OUT/prof -name "*kernel_stats*.csv" | head -1) > $OUT/rollout_bf16_graph_kernel_stats.txt
grep "rollout bf16" $OUT/prof_rollout.log >> $OUT/rollout_bf16_graph_kernel_stats.txt
rm -rf $OUT/prof
timeout -k 10 300 python scripts/time_linear16.py > $OUT/time_linear16.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/time_linear16.py --skip-check --heads >> $OUT/time_linear16.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/time_heads.py > $OUT/time_heads.txt 2>&1
head -30 $OUT/kernel_stats_ppo.txt; cat $OUT/update_timeline.txt; head -12 $OUT/rollout_bf16_graph_kernel_stats.txt; tail -12 $OUT/time_linear16.txt
