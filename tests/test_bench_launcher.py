"""bench.py's own multi-rank launcher (`python bench.py --gpus N` without torchrun) on CPU: BRL_BENCH_FAKE=1 swaps
the process group to gloo and the step to a sleep (no GPU, no compute, no oracle) — what is tested is the launcher and
the timing protocol: N children, RANK / WORLD_SIZE / rendezvous on 127.0.0.1, barrier + max-over-ranks, ONE JSON line
from rank 0 with n_gpus == N and a whole-job value, non-zero exit when WORLD_SIZE and --gpus disagree."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(BRL_BENCH_FAKE="1", **(env_extra or {}))
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = run(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["env_offsets"] == [0, 8192] and out["config"]["parallelism"].startswith("env-shard x2")
    # whole-job value: both shards' macro-steps over the max-over-ranks time
    assert abs(out["value"] - 2 * 8192 * 32 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-6 * out["value"]
    assert out["ms_per_step"] >= 2.0  # the fake step sleeps 2 ms
    assert out["cpu_baseline"] is None and "FAKE" in out["data"]
    # one record per rank (device identity, its own step time): the proof of N distinct devices in an RCCL run
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and len({r["pid"] for r in out["ranks"]}) == 2
    assert max(r["local_ms_per_step"] for r in out["ranks"]) == out["ms_per_step"]


def test_single_rank_default_and_world_size_mismatch():
    r = run(["--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # under a launcher whose world disagrees with --gpus the bench refuses instead of printing an n_gpus: 1 line
    r = run(["--gpus", "4", "--steps", "1"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
