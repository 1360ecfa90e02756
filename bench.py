#!/usr/bin/env python3
"""bench.py — env-steps/s of the fused bridge-bidding rollout on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]           # N > 1: starts N ranks itself (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # or under torchrun (WORLD_SIZE must equal N)
    python bench.py --config ppo [--steps K]                        # secondary: configs[3] phase timing (not the metric)

One "step" = one pass of the hot path over one batch: the T=32-step random-policy rollout of
num_envs=8192 tables (BASELINE.json configs[1]: ONE fused kernel launch writing the full
time-major Transition buffer, auto-reset + DDS reward included), the observation of the
post-rollout state (runner_state's last_obs) and the GAE(lambda) reverse scan over that trajectory — all by the same
launch (brl_rollout_random_gae; BRL_BENCH_FUSED_GAE=0: brl_rollout_random, then brl_gae, as in earlier rounds).
Inputs (table states, LUT) are resident in HBM before the timed region; the Transition buffer rotates over
NBUF = 3 allocations (433 MB > the 256 MB Infinity Cache), so the stores have to reach HBM.  With N > 1 every rank
runs its own 8192-table shard (weak scaling, no data-path collective — SURVEY §8e); `value` is the whole-job
macro-steps/s = N * 8192 * 32 * K / max-over-ranks time.  Prints ONE JSON line on rank 0.

Protocol: ranks lined up -> untimed device warm-up (--device-warmup-ms of the same step: steady-state clocks) -> W untimed
warm-up steps -> barrier + synchronize -> K steps -> synchronize = the rank's time -> barrier -> MAX over ranks.

`roofline` is measured in THIS run: after the timed region, KERNEL_LAUNCHES back-to-back launches of the rollout kernel
alone (same rotating buffers, the C-ABI called directly) between ONE pair of HIP events on the launch stream;
`roofline.plain_fill`: a plain fill of the bytes one launch writes, timed the same way (the box's store ceiling).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_ENVS = 8192
NUM_STEPS = 32
LUT_LEN = 100_000          # ppo.py:128 hash_size
ROW_BYTES = 535            # obs 480 + mask 38 + action 4 + value 4 + reward 4 + log_prob 4 + done 1
OBS_BYTES = 480            # the observation path alone
LAST_ROW_BYTES = 518       # last_obs 480 + its legal mask 38, written once per table by the same launch (not counted)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NBUF = int(os.environ.get("BRL_BENCH_NBUF", "3"))  # Transition buffers in rotation: 3 x 144 MB > 256 MB Infinity Cache
KERNEL_LAUNCHES = 128      # launches between the one event pair of the kernel-only loop
METRIC = "env-steps/sec at num_envs=8192, 32-step rollout, 1/2/4/8 MI355X"
FAKE = os.environ.get("BRL_BENCH_FAKE") == "1"  # launcher self-test (tests/test_bench_launcher.py): gloo, no GPU, no compute


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
    """HBM bytes per rollout launch from the committed rocprofv3 --pmc summary (a SEPARATE run of this command:
    counters cannot be collected inside the timed run), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
        return d.get("rollout_hbm_bytes_per_launch"), d.get("source", "profiles/pmc_traffic.json")
    except Exception:
        return None, None


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    """One child process per GPU (rank r -> cuda:r), rendezvous on 127.0.0.1; the parent never touches the GPU.
    Rank 0's stdout (the ONE JSON line) is passed through; the exit code is the worst child's."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="rollout", choices=["rollout", "ppo"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-warmup-ms", type=float, default=80.0,
                    help="untimed: the same step back to back for this long BEFORE the --warmup steps, so that the timed steps see "
                         "the device's steady-state clocks (a launch takes 26.4 us in the first milliseconds after idle, 25.0 us "
                         "from ~50 ms on; 20 timed steps last 0.5 ms)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # before anything initialises a GPU in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import torch

    dist = None
    # BRL_BENCH_BACKEND=gloo: rehearsal of the N-rank path on a box with fewer GPUs than ranks (ranks share the devices
    # round-robin, the control plane runs over gloo); the driver's runs use RCCL, one GPU per rank
    backend = os.environ.get("BRL_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        if FAKE:
            dist.init_process_group("gloo")
        elif backend == "gloo":
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif not FAKE:
        torch.cuda.set_device(0)
    dev = torch.device("cpu") if FAKE else torch.device("cuda", torch.cuda.current_device())
    ctl_dev = torch.device("cpu") if (FAKE or backend == "gloo") else dev   # where the timing all-reduce lives

    def barrier():
        if dist is not None:
            dist.barrier()
        if not FAKE:
            torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.config == "ppo":
        out = bench_ppo(args, torch, dev, rank, world, barrier, max_over_ranks)
    else:
        out = bench_rollout(args, torch, dev, rank, world, barrier, max_over_ranks)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_rollout(args, torch, dev, rank, world, barrier, max_over_ranks):
    env_offset = rank * NUM_ENVS
    keys, values = synthetic_lut(LUT_LEN, 0)
    kern_ms = float("nan")
    if FAKE:
        def one_step(i):
            time.sleep(0.002)
    else:
        import ctypes as C

        import brl_amd
        from brl_amd import _capi
        from brl_amd.roll_out import alloc_transition

        env = brl_amd.BridgeBidding(lut=(keys, values), device=dev, env_offset=env_offset)
        state = env.init(0, num_envs=NUM_ENVS)
        packed = state.packed
        trajs = [alloc_transition(NUM_STEPS, NUM_ENVS, dev) for _ in range(NBUF)]
        # everything a step touches is allocated once: the production loop (brl_amd/train.py) does the same, and at
        # ~30 us of GPU time per step the host must not spend its time in the allocator
        advs = [torch.empty((NUM_STEPS, NUM_ENVS), dtype=torch.float32, device=dev) for _ in range(NBUF)]
        tgts = [torch.empty((NUM_STEPS, NUM_ENVS), dtype=torch.float32, device=dev) for _ in range(NBUF)]
        last_val = torch.zeros(NUM_ENVS, dtype=torch.float32, device=dev)  # the random policy has no critic
        last_obs = torch.empty((NUM_ENVS, 480), dtype=torch.bool, device=dev)
        last_mask = torch.empty((NUM_ENVS, 38), dtype=torch.bool, device=dev)
        tc = torch.zeros(1, dtype=torch.int64, device=dev)
        ptrs = []
        for tr in trajs:
            p = _capi.TransitionPtrs()
            for name in _capi.TransitionPtrs._names:
                setattr(p, name, getattr(tr, name).data_ptr())
            ptrs.append(p)
        lib, h = _capi.lib(), env._h
        main = torch.cuda.current_stream()
        # BRL_BENCH_OVERLAP=1 (experiment, off by default): GAE of step i on a second stream beside the rollout of step
        # i+1 (it only reads step i's buffer); rollout_done[k] orders GAE after its rollout, gae_done[k] orders the NEXT
        # writer of buffer k after that GAE.  Measured: 38.4 us per step against 31.6 us in one stream — the two
        # cross-stream event edges per step cost more than the 5-us GAE launch they hide.
        overlap = os.environ.get("BRL_BENCH_OVERLAP", "0") == "1"
        side = torch.cuda.Stream(device=dev) if overlap else main
        rollout_done = [torch.cuda.Event() for _ in range(NBUF)]
        gae_done = [torch.cuda.Event() for _ in range(NBUF)]
        gl = float(torch.tensor(1.0 * 0.95, dtype=torch.float32))  # gamma * gae_lambda as ppo.py forms it
        box = {"draw": 0}

        # The step is ONE launch (brl_rollout_random_gae): the rollout kernel also runs calc_gae's reverse scan over the
        # trajectory it has just written (value == 0 for the random policy; last_val is an input) — on its logic wave, which
        # is done ~10 k cycles before the emit waves.
        # BRL_BENCH_FUSED_GAE=0: the two-launch step of earlier rounds (brl_rollout_random, then brl_gae).
        fused_gae = os.environ.get("BRL_BENCH_FUSED_GAE", "1") != "0" and not overlap

        def one_step(i):
            k = i % NBUF
            if fused_gae:
                _capi.check(lib.brl_rollout_random_gae(h, packed.data_ptr(), NUM_ENVS, NUM_STEPS, box["draw"] & 0xFFFFFFFF, 7600.0,
                                                       C.byref(ptrs[k]), last_obs.data_ptr(), last_mask.data_ptr(), tc.data_ptr(),
                                                       last_val.data_ptr(), 1.0, gl, advs[k].data_ptr(), tgts[k].data_ptr(),
                                                       main.cuda_stream))
                box["draw"] += NUM_STEPS
                return
            if overlap:
                main.wait_event(gae_done[k])          # buffer k's previous GAE has read it
            _capi.check(lib.brl_rollout_random(h, packed.data_ptr(), NUM_ENVS, NUM_STEPS, 1, box["draw"] & 0xFFFFFFFF, 7600.0,
                                               C.byref(ptrs[k]), last_obs.data_ptr(), last_mask.data_ptr(), tc.data_ptr(),
                                               main.cuda_stream))
            box["draw"] += NUM_STEPS
            if overlap:
                rollout_done[k].record(main)
                side.wait_event(rollout_done[k])
            tr = trajs[k]
            _capi.check(lib.brl_gae(h, tr.done.data_ptr(), tr.value.data_ptr(), tr.reward.data_ptr(), last_val.data_ptr(),
                                    1.0, gl, NUM_STEPS, NUM_ENVS, advs[k].data_ptr(), tgts[k].data_ptr(), side.cuda_stream))
            if overlap:
                gae_done[k].record(side)

    # ranks finish their set-up (LUT build, allocations) up to seconds apart: line them up BEFORE the device warm-up — and run
    # both kinds of collective once — so that the barrier in front of the timed region finds every rank already there and no
    # GPU sits idle (and clocks down) while it waits for the others
    barrier()
    max_over_ranks(0.0)
    ramp_steps = 0
    if not FAKE and args.device_warmup_ms > 0:
        t_ramp = time.perf_counter() + args.device_warmup_ms * 1e-3
        while time.perf_counter() < t_ramp:  # (host-timed, in blocks: the queue stays a few hundred launches deep at most)
            for i in range(256):
                one_step(ramp_steps + i)
            ramp_steps += 256
            torch.cuda.synchronize()
    for i in range(args.warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    if not FAKE:
        torch.cuda.synchronize()
    local = time.perf_counter() - t0  # this rank's K steps, from the common start to its own last store
    barrier()
    elapsed = max_over_ranks(local)   # the job's K steps = the slowest rank's (the closing barrier's own latency — one RCCL
                                      # all-reduce, ~0.1 ms, a fifth of 20 steps — is not part of anybody's steps)

    if not FAKE:
        # the dominant kernel alone: KERNEL_LAUNCHES launches of k_rollout_fs between ONE HIP-event pair on the
        # launch stream (no per-launch events: a pair costs ~6 us of stream time), through the C-ABI directly
        draw = box["draw"]
        stream = main

        def launch(i):  # the same launch as the step's
            if fused_gae:
                _capi.check(lib.brl_rollout_random_gae(h, packed.data_ptr(), NUM_ENVS, NUM_STEPS, (draw + i * NUM_STEPS) & 0xFFFFFFFF,
                                                       7600.0, C.byref(ptrs[i % NBUF]), last_obs.data_ptr(), last_mask.data_ptr(),
                                                       tc.data_ptr(), last_val.data_ptr(), 1.0, gl, advs[i % NBUF].data_ptr(),
                                                       tgts[i % NBUF].data_ptr(), stream.cuda_stream))
                return
            _capi.check(lib.brl_rollout_random(h, packed.data_ptr(), NUM_ENVS, NUM_STEPS, 1, (draw + i * NUM_STEPS) & 0xFFFFFFFF,
                                               7600.0, C.byref(ptrs[i % NBUF]), last_obs.data_ptr(), last_mask.data_ptr(),
                                               tc.data_ptr(), stream.cuda_stream))
        for i in range(8):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for i in range(KERNEL_LAUNCHES):
            launch(8 + i)
        e1.record(stream)
        torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / KERNEL_LAUNCHES
        # the chip's own store ceiling on THIS box, for scale: a plain fill of the bytes one launch writes, same rotating buffers
        # (untimed w.r.t. the metric; 128 fills between one event pair)
        written = ROW_BYTES * NUM_ENVS * NUM_STEPS + LAST_ROW_BYTES * NUM_ENVS + 128 * NUM_ENVS + (8 * NUM_ENVS * NUM_STEPS if fused_gae else 0)
        fills = [torch.empty(written, dtype=torch.uint8, device=dev) for _ in range(NBUF)]
        for i in range(16):
            fills[i % NBUF].fill_(1)
        torch.cuda.synchronize()
        e0.record(stream)
        for i in range(KERNEL_LAUNCHES):
            fills[i % NBUF].fill_(1)
        e1.record(stream)
        torch.cuda.synchronize()
        fill_ms = e0.elapsed_time(e1) / KERNEL_LAUNCHES
        del fills

    rows = NUM_ENVS * NUM_STEPS
    alg_bytes = ROW_BYTES * rows                      # 535 B per macro-step row x rows per launch
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    achieved_obs = OBS_BYTES * rows / (kern_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic()
    macro_steps = world * rows * args.steps
    out = {
        "metric": METRIC,
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
        "data": "FAKE (launcher self-test, no compute)" if FAKE else "synthetic",
        "config": {"workload": "configs[1]: num_envs=8192 num_steps=32 random-policy rollout + DDS reward (fused "
                               "kernel) + last_obs + GAE scan", "num_envs_per_gpu": NUM_ENVS, "num_steps": NUM_STEPS,
                   "launches_per_step": 2 if (FAKE or os.environ.get("BRL_BENCH_FUSED_GAE", "1") == "0"
                                              or os.environ.get("BRL_BENCH_OVERLAP", "0") == "1") else 1,
                   "lut_len": LUT_LEN, "env_steps_per_macro_step": 1, "transition_buffers_in_rotation": NBUF,
                   "gae": "second stream, beside the next step's rollout" if os.environ.get("BRL_BENCH_OVERLAP", "0") == "1"
                   else ("brl_gae, same stream" if os.environ.get("BRL_BENCH_FUSED_GAE", "1") == "0"
                         else "inside the rollout launch: its logic wave scans the trajectory once the scorer is done (brl_rollout_random_gae)"),
                   "env_offsets": [r * NUM_ENVS for r in range(world)],
                   "device_warmup": f"{ramp_steps} untimed steps ({args.device_warmup_ms:g} ms) before the --warmup steps: steady-state clocks",
                   "parallelism": f"env-shard x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k_rollout_fs", "kernel_ms": kern_ms,
                     "kernel_ms_method": f"{KERNEL_LAUNCHES} back-to-back launches between one HIP-event pair, "
                                         f"{NBUF} rotating 144 MB output buffers (rank 0)",
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "obs_path": {"bytes_per_launch": OBS_BYTES * rows, "achieved": achieved_obs,
                                  "frac": achieved_obs / HBM_PEAK_GBS},
                     "not_counted_bytes_per_launch": LAST_ROW_BYTES * NUM_ENVS + 2 * 128 * NUM_ENVS
                                                     + (8 * rows if (not FAKE and fused_gae) else 0)},  # + advantages / targets
    }
    if not FAKE:
        out["roofline"]["plain_fill"] = {"bytes": written, "ms": fill_ms, "GB/s": written / (fill_ms * 1e-3) / 1e9,
                                         "kernel_over_fill": kern_ms / fill_ms,
                                         "what": "torch fill_ of the bytes one launch writes, same box, same rotation: the device's store ceiling"}
    if rank == 0:
        out["cpu_baseline"] = cpu_baseline(keys, values) if (world == 1 and not args.no_cpu_baseline and not FAKE) else None
    return out


def bench_ppo(args, torch, dev, rank, world, barrier, max_over_ranks):
    """Secondary mode (NOT the BASELINE metric): BASELINE.json configs[3] — the ppo.py iteration at num_envs=8192,
    num_steps=32, minibatch 1024, 10 epochs with the DeepMind MLP: roll_out (4 forwards + 4 env sub-steps per macro-
    step), calc_gae, update_step, timed per phase; GEMM throughput against the dense MFMA peaks."""
    import brl_amd
    from brl_amd.models import make_forward_pass
    from brl_amd.train import DEFAULTS
    from brl_amd.update import make_optimizer, make_update_step

    if os.environ.get("BRL_TUNABLEOP") == "1":  # experiment: let torch pick the fastest rocBLAS / hipBLASLt solution per GEMM shape
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(30)
    cfg = dict(DEFAULTS, num_envs=NUM_ENVS, num_steps=NUM_STEPS, minibatch_size=1024, update_epochs=10,
               inference_dtype=os.environ.get("BRL_INFER_DTYPE", "bf16"), graph_rollout=True)
    cfg["num_minibatches"] = cfg["num_envs"] * cfg["num_steps"] // cfg["minibatch_size"]
    keys, values = synthetic_lut(LUT_LEN, 0)
    env = brl_amd.BridgeBidding(lut=(keys, values), device=dev, env_offset=rank * NUM_ENVS)
    fp = make_forward_pass("relu", "DeepMind")
    params = fp.init(0, device=dev)
    opt_state = make_optimizer(cfg, params)
    roll_out = brl_amd.make_roll_out(cfg, env, fp, fp)
    calc_gae = brl_amd.make_calc_gae(cfg, fp)
    update_step = make_update_step(cfg, fp)
    st = env.init(0, num_envs=NUM_ENVS)
    rs = (params, opt_state, st, st.observation, 0, 0)
    iters = max(1, min(args.steps, 5))
    phases = {"rollout": [], "gae": [], "update": []}

    def one_iter(rs, record):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rs, traj = roll_out(rs, rs[0])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        adv, tgt = calc_gae(rs, traj)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        rs, _ = update_step(rs, traj, adv, tgt)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if record:
            phases["rollout"].append(t1 - t0); phases["gae"].append(t2 - t1); phases["update"].append(t3 - t2)
        return rs

    rs = one_iter(rs, False)  # graph capture, hipBLASLt heuristics
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        rs = one_iter(rs, True)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    med = {k: float(np.median(v)) for k, v in phases.items()}
    rows = NUM_ENVS * NUM_STEPS
    fwd_flop = 2 * 3_677_184                      # 7.354 MFLOP per forward per sample
    roll_flop = 4 * rows * fwd_flop + NUM_ENVS * fwd_flop
    upd_flop = 3 * rows * fwd_flop * cfg["update_epochs"]
    return {
        "metric": "ppo.py iteration macro-steps/sec at num_envs=8192, num_steps=32, minibatch 1024, 10 epochs (secondary, configs[3])",
        "value": world * rows * iters / elapsed, "unit": "macro-steps/s", "n_gpus": world, "steps": iters, "warmup": 1,
        "ms_per_step": elapsed / iters * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"rollout inference {cfg['inference_dtype']}, update fp32", "data": "synthetic",
        "config": {"workload": "configs[3]: roll_out (policy in the loop, competitive) + calc_gae + update_step",
                   "num_envs_per_gpu": NUM_ENVS, "num_steps": NUM_STEPS, "minibatch_size": 1024, "update_epochs": 10,
                   "graph_rollout": True},
        "phases_ms": {k: v * 1e3 for k, v in med.items()},
        "rollout": {"macro_steps_per_s": rows / med["rollout"], "raw_env_steps_per_s": 4 * rows / med["rollout"],
                    "gemm_tflops": roll_flop / med["rollout"] / 1e12,
                    "mfma_peak_tflops": 2500.0 if cfg["inference_dtype"] in ("bf16", "fp16") else 157.3},
        "update": {"gemm_tflops": upd_flop / med["update"] / 1e12, "mfma_peak_tflops": 157.3,
                   "note": "fp32 GEMMs (fwd + 2x bwd), 2560 minibatch steps of 1024 samples, hipGraph-replayed"},
    }


if __name__ == "__main__":
    main()
