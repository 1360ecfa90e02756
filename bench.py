#!/usr/bin/env python3
"""bench.py — env-steps/s of the fused bridge-bidding rollout on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: the T=32-step random-policy rollout of
num_envs=8192 tables (BASELINE.json configs[1]: ONE fused kernel launch writing the full
time-major Transition buffer, auto-reset + DDS reward included), the observation of the
post-rollout state (runner_state's last_obs, written by the same launch) and the GAE(lambda) reverse scan.  Inputs (table
states, LUT) are resident in HBM before the timed region.  With N > 1 every rank runs its own
8192-table shard (weak scaling, no data-path collective — SURVEY §8e); `value` is the
whole-job macro-steps/s = N * 8192 * 32 * K / max-over-ranks time.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_ENVS = 8192
NUM_STEPS = 32
LUT_LEN = 100_000          # ppo.py:128 hash_size
ROW_BYTES = 535            # obs 480 + mask 38 + action 4 + value 4 + reward 4 + log_prob 4 + done 1
LAST_ROW_BYTES = 518       # last_obs 480 + its legal mask 38, written once per table by the same launch
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synthetic_lut(n: int, seed: int = 0):
    """numpy default_rng(0) shuffles of 52 cards, tricks uniform 0..13 (pgx packing)."""
    rng = np.random.default_rng(seed)
    owner = rng.permuted(np.tile(np.repeat(np.arange(4, dtype=np.int64), 13), (n, 1)), axis=1)  # [n,52] by card id
    w = 4 ** np.arange(12, -1, -1, dtype=np.int64)
    keys = (owner.reshape(n, 4, 13) * w).sum(-1).astype(np.int32)
    tricks = rng.integers(0, 14, size=(n, 4, 5), dtype=np.int64)
    h = 16 ** np.arange(4, -1, -1, dtype=np.int64)
    values = (tricks * h).sum(-1).astype(np.int32)
    return keys, values


def cpu_baseline(keys, values, budget_s: float = 12.0):
    """The CPU oracle (oracle/bridge_oracle.c, OpenMP over envs) on the same workload, timed on
    this box's host cores.  A restatement ("port"), NOT the JAX reference — see BASELINE.md §2."""
    try:
        from oracle import Oracle
        orc = Oracle(keys, values)
        threads = len(os.sched_getaffinity(0))
        st = orc.init_random(NUM_ENVS, seed=0)
        orc.rollout_random(st, NUM_STEPS, seed=0)  # warm-up
        times = []
        t_end = time.perf_counter() + budget_s
        draw = NUM_STEPS
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 200):
            t0 = time.perf_counter()
            orc.rollout_random(st, NUM_STEPS, seed=0, draw_base=draw)
            times.append(time.perf_counter() - t0)
            draw += NUM_STEPS
        med = float(np.median(times))
        return {"value": NUM_ENVS * NUM_STEPS / med, "unit": "macro-steps/s", "cores": threads, "kind": "port",
                "sample": f"{len(times)} rollouts of num_envs={NUM_ENVS} x num_steps={NUM_STEPS} (random policy, "
                          f"auto-reset, full Transition stored), median; C oracle with OpenMP over envs"}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "macro-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


def pmc_traffic():
    """HBM bytes per rollout launch from the committed rocprofv3 --pmc summary, if any."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path)).get("rollout_hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import brl_amd
    from brl_amd.gae import gae_scan
    from brl_amd.roll_out import alloc_transition

    keys, values = synthetic_lut(LUT_LEN, 0)
    env = brl_amd.BridgeBidding(lut=(keys, values), device=dev, env_offset=rank * NUM_ENVS)
    cfg = {"num_steps": NUM_STEPS, "game_mode": "normal", "reward_scale": 7600, "return_last_obs": True}
    roll = brl_amd.make_random_roll_out(cfg, env)
    state = env.init(0, num_envs=NUM_ENVS)
    traj = alloc_transition(NUM_STEPS, NUM_ENVS, dev)
    last_val = torch.zeros(NUM_ENVS, dtype=torch.float32, device=dev)  # the random policy has no critic
    rs = (None, None, state, None, 0, 0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(rs, ev=None):
        if ev is not None:
            ev[0].record()
        rs, tb = roll(rs, out=traj)
        if ev is not None:
            ev[1].record()
        adv, tgt = gae_scan(env, tb.done, tb.value, tb.reward, last_val, 1.0, 0.95)
        return rs

    for _ in range(args.warmup):
        rs = one_step(rs)
    # HIP events around the rollout launch of every EV_EVERY-th timed step: an event pair costs ~6 us of stream
    # time on this stack (scripts/graph_probe.py: 39.1 us per step without, 45.3 with a pair on every step), so
    # bracketing every launch would measure the instrumentation.  The sampled launches are inside the timed region.
    EV_EVERY = 8
    events = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for i in range(0, args.steps, EV_EVERY)}
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        rs = one_step(rs, events.get(i))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: k_rollout_random; HIP events on the launch stream, inside the timed region,
    # bracketing exactly that one launch.
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events.values()])) if events else float("nan")
    alg_bytes = ROW_BYTES * NUM_ENVS * NUM_STEPS + LAST_ROW_BYTES * NUM_ENVS
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    macro_steps = world * NUM_ENVS * NUM_STEPS * args.steps
    out = {
        "metric": "env-steps/sec at num_envs=8192, 32-step rollout, 1/2/4/8 MI355X",
        "value": macro_steps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "configs[1]: num_envs=8192 num_steps=32 random-policy rollout + DDS reward (fused "
                               "kernel) + last_obs + GAE scan", "num_envs_per_gpu": NUM_ENVS, "num_steps": NUM_STEPS,
                   "lut_len": LUT_LEN, "env_steps_per_macro_step": 1, "tables_per_wave": os.environ.get("BRL_TABLES_PER_WAVE", "4"),
                   "parallelism": f"env-shard x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
                     "kernel": "k_rollout_ws<%s>" % os.environ.get("BRL_ROLLOUT_WS", "32x12").replace("x", ","), "kernel_ms": kern_ms, "kernel_ms_samples": len(events), "algorithmic_bytes_per_launch": alg_bytes},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(keys, values)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
