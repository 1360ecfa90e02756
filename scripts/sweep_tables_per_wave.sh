#!/bin/bash
# BRL_TABLES_PER_WAVE sweep of the per-step kernels on the policy-in-the-loop path (scripts/bench_policy.py numbers per K)
for k in 1 2 4 8; do
  BRL_TABLES_PER_WAVE=$k timeout 300 python scripts/bench_policy.py 2>/dev/null | tail -1 > /tmp/bp_$k.json
  python3 - "$k" <<'PY'
import json, sys
k = sys.argv[1]
d = json.load(open(f"/tmp/bp_{k}.json"))
print("K=%s rollout fp32 %.2f ms  bf16 %.2f  bf16+graph %.2f  duplicate eval %.2f ms" % (k, d["rollout_fp32_ms"], d["rollout_bf16_ms"], d["rollout_bf16_graph_ms"], d["duplicate_eval_8192_boards_ms"]))
PY
done
