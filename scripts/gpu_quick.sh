#!/bin/bash
# Quick GPU session: parity tests, smoke, bench (no profiling).  usage: bash scripts/gpu_quick.sh [tag] [pytest -k expr]
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
if [ -n "$2" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu -k "$2" 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
else
  timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
fi
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee $OUT/bench.json
