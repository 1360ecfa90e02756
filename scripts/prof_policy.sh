#!/bin/bash
# Policy-in-the-loop path (BASELINE configs[2]/[3]): phase timing + rocprofv3 kernel trace of `bench.py --config ppo`.
TAG=${1:-r02_policy}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for dt in bf16 fp32; do
  BRL_INFER_DTYPE=$dt timeout 900 python bench.py --config ppo --steps 3 2>/dev/null | tail -1 > $OUT/bench_ppo_$dt.json
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --config ppo --steps 2 > $OUT/prof.log 2>&1
F=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1)
python3 scripts/short_stats.py $F > $OUT/kernel_stats_ppo.txt
rm -rf $OUT/prof
head -45 $OUT/kernel_stats_ppo.txt
python3 - <<PY
import json
for dt in ("bf16","fp32"):
    d=json.load(open("$OUT/bench_ppo_%s.json"%dt)); print(dt, d["phases_ms"], "update GEMM TFLOP/s %.1f"%d["update"]["gemm_tflops"], "rollout GEMM TFLOP/s %.1f"%d["rollout"]["gemm_tflops"], "iteration macro-steps/s %.0f"%d["value"])
PY
