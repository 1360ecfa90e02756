#!/bin/bash
# The day a multi-GPU MI355X node is available: everything that has never run at world > 1, in one command, into profiles/node/.
#   bash scripts/node_day.sh [out dir]            on a node (uses every visible GPU, at most 8)
#   NODE_DRY=1 bash scripts/node_day.sh [out]     on a one-GPU box: the same script at world 1 (BRL_FORCE_DIST=1: every collective really
#                                                 goes through RCCL with one peer), reduced sizes for the ppo.py legs — the dry run of
#                                                 tests/test_multi_gpu_rccl.py::test_node_day_script_dry_run
# Steps (each leaves a file; a failing step is recorded and the script goes on):
#   1  the RCCL tests that SKIP below 2 GPUs (tests/test_multi_gpu_rccl.py, the two-rank tests of tests/test_gpu_parity.py)
#   2  bench.py --gpus 1, 2, 4, 8 (the BASELINE metric, weak scaling: rollout shards, no data-path collective)
#   3  bench.py --gpus N --config ppo with the gradient step "flat" and "sharded" (configs[4]: RCCL nodes inside the step's hipGraph)
#   4  scripts/allreduce_graph_probe.py: the 14.7 MB all-reduce and the sharded form's buckets inside a graph — the number DESIGN §7's
#      0.42 ms-per-step estimate rests on
OUT=${1:-profiles/node}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
PY=${PYTHON:-python}
if [ -n "$NODE_DRY" ]; then
  N=1; export BRL_FORCE_DIST=1; export BRL_BENCH_PPO_ENVS=${BRL_BENCH_PPO_ENVS:-1024}; export BRL_BENCH_PPO_EPOCHS=${BRL_BENCH_PPO_EPOCHS:-2}
  GPUS_LIST="1"
else
  N=$($PY -c "import torch; print(min(8, torch.cuda.device_count()))")
  GPUS_LIST=$(for g in 1 2 4 8; do [ $g -le $N ] && echo -n "$g "; done)
fi
echo "node_day: $N GPU(s), dry=${NODE_DRY:-0}, out=$OUT" | tee "$OUT/summary.txt"
step() { echo "== $1" | tee -a "$OUT/summary.txt"; }

step "1 RCCL tests"
if [ -n "$NODE_DRY" ]; then K="world_1 or capture_group"; else K="rccl or two_ranks or ranks"; fi
timeout -k 10 1500 $PY -m pytest tests/test_multi_gpu_rccl.py tests/test_gpu_parity.py -q -m gpu -k "$K" --deselect tests/test_multi_gpu_rccl.py::test_node_day_script_dry_run > "$OUT/pytest_rccl.txt" 2>&1
tail -3 "$OUT/pytest_rccl.txt" | tee -a "$OUT/summary.txt"

step "2 bench.py --gpus $GPUS_LIST"
for g in $GPUS_LIST; do
  EXTRA=""; [ $g -gt 1 ] && EXTRA="--no-cpu-baseline"
  timeout -k 10 900 $PY bench.py --gpus $g --steps 200 --warmup 20 $EXTRA 2> "$OUT/bench_gpus$g.err" | tail -1 > "$OUT/bench_gpus$g.json"
  $PY -c "import json,sys; d=json.load(open('$OUT/bench_gpus$g.json')); print('  gpus', d['n_gpus'], 'value %.4g' % d['value'], d['unit'], 'ms/step %.4f' % d['ms_per_step'], d.get('ranks_backend'))" 2>&1 | tee -a "$OUT/summary.txt"
done

step "3 bench.py --gpus $N --config ppo (flat, sharded)"
for mode in flat sharded; do
  BRL_GRAD_ALLREDUCE=$mode timeout -k 10 1200 $PY bench.py --gpus $N --config ppo --steps 3 2> "$OUT/bench_ppo_${mode}_gpus$N.err" | tail -1 > "$OUT/bench_ppo_${mode}_gpus$N.json"
  $PY -c "import json; d=json.load(open('$OUT/bench_ppo_${mode}_gpus$N.json')); print('  $mode: value %.4g' % d['value'], d['unit'], 'phases_ms', {k: round(v, 2) for k, v in d['phases_ms'].items()}, 'in graph', d['config']['collectives_inside_the_graph'])" 2>&1 | tee -a "$OUT/summary.txt"
done

step "4 gradient collectives inside a graph"
if [ $N -gt 1 ]; then
  timeout -k 10 600 $PY -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29577 scripts/allreduce_graph_probe.py "$OUT/allreduce_graph_probe.json" > "$OUT/allreduce_graph_probe.log" 2>&1
else
  timeout -k 10 600 $PY scripts/allreduce_graph_probe.py "$OUT/allreduce_graph_probe.json" > "$OUT/allreduce_graph_probe.log" 2>&1
fi
$PY -c "import json; d=json.load(open('$OUT/allreduce_graph_probe.json')); print('  world', d['world'], {k: round(v, 1) for k, v in d['us_per_collective'].items()}, 'busbw GB/s %.1f' % d['all_reduce_flat_busbw_GBps'])" 2>&1 | tee -a "$OUT/summary.txt"
echo "node_day: done" | tee -a "$OUT/summary.txt"
