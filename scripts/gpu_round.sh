#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes (separate, kernel-trace only).
# usage (from the repo root on the GPU box): bash scripts/gpu_round.sh [tag]
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/summary.txt
# (the whole log is kept, and the test that was running: a crash of the process must say where)
rm -f $OUT/pytest_trace.txt
BRL_TEST_TRACE=$OUT/pytest_trace.txt timeout -k 10 1500 python -X faulthandler -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_full.txt 2>&1
tail -8 $OUT/pytest_gpu_full.txt | tee $OUT/pytest_gpu.txt
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 | tee $OUT/bench.json
echo "== rocprofv3 kernel trace" | tee -a $OUT/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/prof_trace.log 2>&1
find $OUT/prof_trace -name "*kernel_stats*.csv" | head -1 | xargs -r cat | head -20 | tee $OUT/kernel_stats.csv
rm -rf $OUT/prof_trace
echo "== rocprofv3 pmc WRITE_SIZE / FETCH_SIZE (separate passes)" | tee -a $OUT/summary.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_pmc_w -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/prof_pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc_r -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/prof_pmc_r.log 2>&1
python3 scripts/summarize_pmc.py $OUT 2>&1 | tee $OUT/pmc_summary.txt
rm -rf $OUT/prof_pmc_w $OUT/prof_pmc_r
echo "== instruction mix (SQ counters, two passes)"
bash scripts/pmc_mix.sh > $OUT/pmc_mix.txt 2>&1; tail -30 $OUT/pmc_mix.txt
rm -rf gpurun_out/pmc_mix
ls -la $OUT
