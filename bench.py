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


# fp32 GEMM FLOPs one sample costs in the PPO update AS EXECUTED (DeepMind MLP 480 -> 4 x 1024 -> 38 + 1): forward 2 x 3 677 184,
# weight gradients the same again, input gradients for everything but layer 0 (480 x 1024): 21.08 MFLOP, not 3 x forward = 22.06
UPDATE_FLOP_PER_SAMPLE = 2 * (3 * 3_677_184 - 480 * 1024)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """-> (CPUs this process may use at once, where that figure comes from).  The affinity mask of a container lists every core
    of the host while the box's CPU SHARE (cgroup quota) is a fraction of it: an OpenMP team larger than the quota bursts for
    a few milliseconds and is then throttled for the rest of every 100 ms period.  cgroup v2 `cpu.max`, then cgroup v1
    `cpu.cfs_quota_us / cpu.cfs_period_us` (own cgroup first, then the mount's root); (None, reason) when no quota is set."""
    paths = []
    try:
        for line in open("/proc/self/cgroup"):
            _, ctrl, path = line.rstrip("\n").split(":", 2)
            if ctrl == "":
                paths += [("v2", os.path.join("/sys/fs/cgroup", path.lstrip("/"))), ("v2", "/sys/fs/cgroup")]
            elif "cpu" in ctrl.split(","):
                paths += [("v1", os.path.join("/sys/fs/cgroup", ctrl, path.lstrip("/"))), ("v1", os.path.join("/sys/fs/cgroup", ctrl)),
                          ("v1", "/sys/fs/cgroup/cpu")]
    except OSError:
        pass
    paths += [("v2", "/sys/fs/cgroup"), ("v1", "/sys/fs/cgroup/cpu"), ("v1", "/sys/fs/cgroup/cpu,cpuacct")]
    best = None
    for kind, d in paths:
        try:
            if kind == "v2":
                q, p = open(os.path.join(d, "cpu.max")).read().split()
                if q == "max":
                    continue
                cpus, src = float(q) / float(p), os.path.join(d, "cpu.max")
            else:
                q = float(open(os.path.join(d, "cpu.cfs_quota_us")).read())
                p = float(open(os.path.join(d, "cpu.cfs_period_us")).read())
                if q <= 0:
                    continue
                cpus, src = q / p, os.path.join(d, "cpu.cfs_quota_us")
        except (OSError, ValueError):
            continue
        if best is None or cpus < best[0]:
            best = (cpus, src)
    return best if best is not None else (None, "no cgroup CPU quota found: candidates bounded by the affinity mask")


def _sustained(call, min_s: float, min_calls: int, max_calls: int = 2000):
    """-> (median s per call, calls, CPUs actually obtained = process CPU time / wall time) over a run of at least `min_s`
    seconds AND `min_calls` calls: long enough for a cgroup quota to throttle an over-sized team (a two-call minimum measures
    the unthrottled burst: round 5's driver record picked 128 threads that way and then collapsed to 3.3 M)."""
    ts = []
    c0, w0 = os.times(), time.perf_counter()
    while len(ts) < min_calls or (time.perf_counter() - w0 < min_s and len(ts) < max_calls):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    c1, w1 = os.times(), time.perf_counter()
    cpu = (c1.user - c0.user) + (c1.system - c0.system)
    return float(np.median(ts)), len(ts), cpu / max(w1 - w0, 1e-9)


def cpu_baseline(keys, values, budget_s: float = 12.0):
    """the CPU oracle (oracle/bridge_oracle.c, OpenMP over envs) rebuilt ON THIS BOX with
    `-O3 -march=native -fopenmp` for this leg (the parity tests keep their own -O2 -ffp-contract=off build), timed on
    this box's host cores: >= 3 warm-up calls, median of >= 10 timed calls, wall clock around a synchronous call.
    A restatement ("port"), NOT the JAX reference — see BASELINE.md §2.
    Team size: every candidate (bounded by 4 x the box's CPU quota where one is set, else by the affinity mask) is judged on a
    SUSTAINED run (>= 0.5 s and >= 10 rollouts, median); the full protocol then runs on the best two and the better one is
    reported; `sweep` carries the whole table, `suspect` is set when configs[1] runs below half of configs[0]'s rate."""
    try:
        from oracle import Oracle
        from oracle.binding import build_native
        lib_file, flags = build_native()
        orc = Oracle(keys, values, lib_file=lib_file)
        import ctypes
        import math
        gomp = ctypes.CDLL("libgomp.so.1")   # the runtime the oracle is linked against: one instance per process
        affinity = len(os.sched_getaffinity(0))
        quota, quota_src = cpu_quota()
        # BRL_CPU_TEAM_LIMIT: an explicit bound for experiments (the whole sweep is reported either way)
        # (a team up to 4 x the quota still gains — the waits of one thread are another's turn: 64 threads on a 16-CPU quota ran
        #  10.9 ms against 14.6 with 16 — beyond it the cgroup throttles the whole team: 87 ms with 128, 189 with 256;
        #  profiles/r06/r06a_cgroup_and_cpu_sweep.txt)
        limit = int(os.environ.get("BRL_CPU_TEAM_LIMIT", 0)) or (min(affinity, max(1, 4 * math.ceil(quota))) if quota else affinity)
        st = orc.init_random(NUM_ENVS, seed=0)
        box = {"draw": 0}

        def one(state=st):
            orc.rollout_random(state, NUM_STEPS, seed=0, draw_base=box["draw"])
            box["draw"] += NUM_STEPS

        sweep = []
        for cand in sorted({c for c in (4, 8, 16, 32, 64, 128, limit) if c <= limit}):
            gomp.omp_set_num_threads(cand)
            one()                                                    # the team's threads exist before the clock starts
            med, calls, got = _sustained(one, 0.5, 10, 400)
            sweep.append({"threads": cand, "median_ms": med * 1e3, "rollouts": calls, "cpus_obtained": round(got, 1)})
        ranked = sorted(sweep, key=lambda r: r["median_ms"])
        finals = []
        for r in ranked[:2]:                                         # the full protocol on the best two
            gomp.omp_set_num_threads(r["threads"])
            for _ in range(3):  # warm-up
                one()
            med, calls, got = _sustained(one, max(1.0, (budget_s - 6.0) / 2), 10, 200)
            finals.append({"threads": r["threads"], "median_ms": med * 1e3, "rollouts": calls, "cpus_obtained": round(got, 1)})
        best = min(finals, key=lambda r: r["median_ms"])
        threads, med, n_timed = best["threads"], best["median_ms"] * 1e-3, best["rollouts"]
        # BASELINE.md §3 lists both sizes: configs[0] (num_envs=128, the reference's own CPU-runnable case) beside configs[1]
        try:
            n0 = 128
            st0 = orc.init_random(n0, seed=0)
            sweep0 = []
            for cand in sorted({c for c in (1, 2, 4, 8, 16, 32) if c <= limit}):   # 128 tables: a small team wins
                gomp.omp_set_num_threads(cand)
                one(st0)
                m0, c0, _ = _sustained(lambda: one(st0), 0.1, 10, 2000)
                sweep0.append((m0, cand))
            t0_best = min(sweep0)[1]
            gomp.omp_set_num_threads(t0_best)
            med0, calls0, _ = _sustained(lambda: one(st0), 1.5, 10, 5000)
            config0 = {"value": n0 * NUM_STEPS / med0, "unit": "macro-steps/s", "cores": t0_best, "num_envs": n0,
                       "num_steps": NUM_STEPS, "ms_per_rollout": med0 * 1e3,
                       "sweep": [{"threads": c, "median_ms": m * 1e3} for m, c in sweep0],
                       "sample": f"configs[0]: {calls0} rollouts of num_envs={n0} x num_steps={NUM_STEPS}, median, same library"}
        except Exception as e:
            config0 = {"value": None, "sample": f"failed: {e}"}
        gomp.omp_set_num_threads(threads)
        value = NUM_ENVS * NUM_STEPS / med
        # a 64-fold larger batch on a larger team must not run BELOW half the rate of 128 tables on a handful of threads: if it
        # does, the team was throttled / over-subscribed on this box and the number is not a baseline
        suspect = bool(config0.get("value")) and value < 0.5 * config0["value"]
        return {"value": value, "unit": "macro-steps/s", "cores": threads, "cpus_obtained": best["cpus_obtained"], "kind": "port", "config0": config0,
                "cpu_model": cpu_model(), "flags": flags, "cpus_in_affinity_mask": affinity,
                "cpu_quota": quota, "cpu_quota_source": quota_src, "team_limit": limit,
                "sweep": sweep, "finals": finals, "suspect": suspect, "ms_per_rollout": med * 1e3,
                "raw_env_steps_per_s": value,   # configs[1]: one env.step per macro-step
                "sample": f"3 warm-ups, then {n_timed} rollouts of num_envs={NUM_ENVS} x num_steps={NUM_STEPS} (random policy, "
                          f"auto-reset, full Transition stored), median; team size = the better of the two best candidates of a "
                          f"sustained sweep (`sweep`); C oracle (a checker: it rebuilds each observation "
                          f"from the whole call history) with OpenMP over envs; CPU restatement (this repo), not the JAX reference"}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "macro-steps/s", "cores": 0, "kind": "port", "cpu_model": cpu_model(), "flags": None,
                "sample": f"failed: {e!r}"}


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


def device_identity(torch, dev):
    """What distinguishes this rank's GPU from the others': PCI bus id (domain:bus:device.function) and uuid where the
    runtime exposes them."""
    out = {"device_index": dev.index, "name": None, "pci_bus_id": None, "uuid": None}
    if dev.type != "cuda":   # launcher self-test on CPU
        return out
    try:
        pr = torch.cuda.get_device_properties(dev)
        out["name"] = pr.name
        if hasattr(pr, "pci_bus_id"):
            out["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
        if getattr(pr, "uuid", None) is not None:
            out["uuid"] = str(pr.uuid)
    except Exception as e:  # identity is evidence, never a reason to lose the measurement
        out["error"] = repr(e)
    if out["pci_bus_id"] is None:
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(32)
            if hip.hipDeviceGetPCIBusId(buf, 32, int(dev.index)) == 0:
                out["pci_bus_id"] = buf.value.decode()
        except Exception:
            pass
    return out


def gather_ranks(torch, dist, rank, world, dev, local_s, kern_ms, steps, backend):
    """-> (list of one record per rank, error string or None).  With RCCL (one GPU per rank) the records must name `world`
    DISTINCT devices; a gloo rehearsal (ranks sharing devices) is labelled as such."""
    rec = dict(device_identity(torch, dev), rank=rank, local_ms_per_step=local_s / steps * 1e3,
               kernel_ms=None if kern_ms != kern_ms else kern_ms, host=socket.gethostname(), pid=os.getpid())
    if dist is None:
        return [rec], None
    recs = [None] * world
    dist.all_gather_object(recs, rec)
    err = None
    if backend == "nccl":
        ids = [r["pci_bus_id"] or r["uuid"] or f"index{r['device_index']}" for r in recs]
        if len(set(ids)) != world:
            err = f"{world} ranks but only {len(set(ids))} distinct GPUs: {ids}"
    return recs, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="rollout", choices=["rollout", "ppo"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip `secondary` (configs[2] duplicate evaluation and configs[3] ppo.py phases with the MLP in the "
                         "loop, ~30 s, N=1 only) and `long_run`")
    ap.add_argument("--secondary-budget-s", type=float, default=60.0,
                    help="time budget of the `secondary` legs (N=1 only, after the metric's timed region): a leg whose estimated "
                         "cost no longer fits is DROPPED and listed in secondary.dropped; 0 = no limit")
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
    # BRL_FORCE_DIST=1 at N = 1: the N-rank control flow (barrier, MAX over ranks, the per-rank records gathered) really over RCCL
    # with one peer — a rehearsal of what the driver's N > 1 runs execute, on a one-GPU box (tests/test_multi_gpu_rccl.py)
    forced = world == 1 and os.environ.get("BRL_FORCE_DIST") == "1" and not FAKE
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or forced:
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

    rc = 0
    if args.config == "ppo":
        out = bench_ppo(args, torch, dev, rank, world, barrier, max_over_ranks)
    else:
        out = bench_rollout(args, torch, dev, rank, world, barrier, max_over_ranks,
                            dist=dist, backend="fake" if FAKE else (backend if (world > 1 or forced) else "none"))
        if out.get("ranks_error"):
            print("bench.py: " + out["ranks_error"], file=sys.stderr)
            rc = 3
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


def bench_rollout(args, torch, dev, rank, world, barrier, max_over_ranks, dist=None, backend="none"):
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

    long_run = python_api = None
    if not FAKE and world == 1 and not args.no_secondary:
        # self-check of `value` on a region long enough for any outside observer (the default K = 20 steps last 0.5 ms):
        # 2000 steps of the same one_step, host-timed the same way
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2000):
            one_step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        long_run = {"steps": 2000, "ms_per_step": dt / 2000 * 1e3, "value": NUM_ENVS * NUM_STEPS * 2000 / dt,
                    "what": "the same step, 2000 times back to back after the timed region (same host clock protocol)"}
        # the same step through the PUBLIC Python surface (brl_amd.make_random_roll_out_with_gae with caller-owned output
        # buffers), not the raw C-ABI call the metric times: what a user of the host mirror gets per call
        roll = brl_amd.make_random_roll_out_with_gae({"num_steps": NUM_STEPS, "gamma": 1.0, "gae_lambda": 0.95}, env)
        rs = (None, None, state, None, tc, box["draw"])
        bufs = [dict(out=trajs[k], out_adv=advs[k], out_tgt=tgts[k], out_last=(last_obs, last_mask)) for k in range(NBUF)]
        for i in range(20):
            rs = roll(rs, last_val, **bufs[i % NBUF])[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(500):
            rs = roll(rs, last_val, **bufs[i % NBUF])[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        python_api = {"steps": 500, "ms_per_step": dt / 500 * 1e3, "value": NUM_ENVS * NUM_STEPS * 500 / dt,
                      "what": "brl_amd.make_random_roll_out_with_gae(out=...) per call: ctypes struct + State wrapper built per "
                              "call on the host (host-bound above ~25 us per call); `value` times the C-ABI entry point itself"}

    ranks, ranks_error = gather_ranks(torch, dist, rank, world, dev, local, kern_ms, args.steps, backend)
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
    out["ranks"] = ranks                      # one record per rank: device index, PCI bus id / uuid, its own step and kernel time
    out["ranks_backend"] = {"nccl": "RCCL, one GPU per rank", "gloo": "gloo rehearsal: ranks may share devices",
                            "fake": "launcher self-test", "none": "single process"}[backend]
    if ranks_error:
        out["ranks_error"] = ranks_error      # (main() exits non-zero)
    if long_run is not None:
        out["long_run"] = long_run
        out["python_api"] = python_api
    if rank == 0:
        out["cpu_baseline"] = cpu_baseline(keys, values) if (world == 1 and not args.no_cpu_baseline and not FAKE) else None
        if world == 1 and not FAKE and not args.no_secondary:
            del trajs, advs, tgts, ptrs   # 433 MB + of rotating buffers: not needed by the policy path
            try:
                out["secondary"] = bench_secondary(torch, dev, args.secondary_budget_s)
            except Exception as e:  # secondary numbers never cost the metric line
                out["secondary"] = {"error": repr(e)}
    return out


# estimated cost (s) of each secondary leg on an MI355X box, first call included (graph capture, library heuristics): what
# the budget check uses BEFORE a leg starts; measured values are reported beside them (secondary.legs_s)
LEG_ESTIMATE_S = {"config2": 4.0, "config3_rollout_fp32": 4.0, "config3_calc_gae": 1.0, "config3_update": 8.0,
                  "config4_rehearsal": 8.0, "config3_evaluators": 8.0, "config3_fair": 12.0,
                  "config3_rollout_bf16": 5.0, "config3_rollout_fp32_bf16x3": 4.0}


def bench_secondary(torch, dev, budget_s: float = 60.0):
    """NOT the metric: the policy-in-the-loop phases the reference times every iteration (ppo.py:466-488) at BASELINE.json
    configs[2] / configs[3] sizes, measured after the metric's timed region in the same process so that the driver's
    record carries them.  fp32 is the reference's precision; the bf16 rollout is an opt-in NARROWER than the reference and
    is labelled as such.  GEMM FLOPs are algorithmic (7.354 MFLOP per forward per sample).
    Legs run in priority order under `budget_s`: a leg whose estimate (LEG_ESTIMATE_S) no longer fits is dropped and named in
    `dropped`; `legs_s` = what each leg took."""
    import brl_amd
    from brl_amd.evaluation import make_simple_duplicate_evaluate
    from brl_amd.models import make_forward_pass
    from brl_amd.train import DEFAULTS
    from brl_amd.update import make_optimizer, make_update_step

    def timed(fn, reps):
        fn()                                           # capture / heuristics / allocator
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), r

    t_begin = time.perf_counter()
    legs_s, dropped = {}, []

    def leg(name, fn, needs=()):
        """runs fn() unless the budget is spent or a leg it needs did not run; a failing leg costs its own numbers only"""
        if any(n not in legs_s or n in failed for n in needs):
            dropped.append({"leg": name, "why": "needs " + ", ".join(needs)})
            return None
        used = time.perf_counter() - t_begin
        if budget_s > 0 and used + LEG_ESTIMATE_S[name] > budget_s:
            dropped.append({"leg": name, "why": f"budget: {used:.1f} s used + {LEG_ESTIMATE_S[name]:g} s estimated > {budget_s:g} s"})
            return None
        t0 = time.perf_counter()
        try:
            return fn()
        except Exception as e:   # (secondary: never in the way of the line)
            failed[name] = repr(e)
            return None
        finally:
            torch.cuda.synchronize()
            legs_s[name] = round(time.perf_counter() - t0, 2)

    failed = {}
    keys, values = synthetic_lut(LUT_LEN, 0)
    fp = make_forward_pass("relu", "DeepMind")
    rows = NUM_ENVS * NUM_STEPS
    fwd_flop = 2 * 3_677_184
    out = {"note": "secondary numbers (not the metric): same process, after the timed region; medians of host-timed, "
                   "synchronised calls; random-init DeepMind MLPs (480 -> 4 x 1024 -> 38 + 1)",
           "budget_s": budget_s}
    env = brl_amd.BridgeBidding(lut=(keys, values), device=dev)
    team1 = fp.init(0, device=dev)
    phases = {}
    cfg32 = dict(DEFAULTS, num_envs=NUM_ENVS, num_steps=NUM_STEPS, minibatch_size=1024, update_epochs=10,
                 inference_dtype=None, graph_rollout=True)
    cfg32["num_minibatches"] = rows // cfg32["minibatch_size"]
    nmb = cfg32["update_epochs"] * cfg32["num_minibatches"]
    keep = {}

    # ---- configs[3]: one ppo.py iteration = roll_out + calc_gae + 10-epoch update_step (ppo.py:466-479)
    def rollout_leg(label, dt, gemm=None):
        cfg = dict(cfg32, inference_dtype=dt, inference_gemm=gemm)
        roll_out = brl_amd.make_roll_out(cfg, env, fp, fp)
        st = env.init(0, num_envs=NUM_ENVS)
        box = {"rs": (team1, None, st, st.observation, 0, 0)}

        def do_roll():
            box["rs"], box["traj"] = roll_out(box["rs"], team1)
            return None
        t_roll, _ = timed(do_roll, 3)
        flop = 4 * rows * fwd_flop
        x3 = gemm == "bf16x3"
        peak = 157.3 if (dt is None and not x3) else 2500.0
        # (bf16x3: six bf16 products per fp32 product — three in the first layer, whose 0/1 input is one plane: what the matrix pipe
        #  executes, against the bf16 peak)
        l0 = 2 * 480 * 1024 / fwd_flop
        flop_exec = (6 * (1 - l0) + 3 * l0) * flop if x3 else flop
        phases["rollout_" + label] = {
            "ms": t_roll * 1e3, "macro_steps_per_s": rows / t_roll, "raw_env_steps_per_s": 4 * rows / t_roll,
            "gemm_tflops": flop / t_roll / 1e12, "mfma_peak_tflops": peak,
            "mfma_frac": flop_exec / t_roll / 1e12 / peak,
            "dtype": ("fp32 inference (the reference's precision), every layer on the library's exact-fp32 GEMM (inference_gemm = 'library')" if not x3 else
                      "fp32 inference, hidden layers as bf16x3 products (fp32 operands as three exact bf16 pieces, six bf16 MFMA products, "
                      "fp32 accumulation: 0.07-0.44 x the exact kernel's error vs float64) on brl_linear_x3p — operands pre-split into "
                      "planes: the weights once per parameter version, an activation by the layer that produces it, the 0/1 "
                      "observation as one bf16 plane (BRL_INFERENCE_PLANES=0: brl_mlp_gemm_x3, the split in registers); "
                      "inference_gemm = 'bf16x3', brl_amd's default") if dt is None else
                     "bf16 inference: NARROWER than the reference's fp32 — opt-in (inference_dtype), never the default",
            "how": "hipGraph-replayed macro-steps: 4 forwards + 4 brl_policy_step_ex launches each (competitive mode)"
                   + ("" if dt is None else "; hidden layers on the library's own bf16 kernel (brl_linear_act), the heads' share inside "
                      "the last layer's launch (brl_linear_act_heads), summed by the sub-step launch")}
        if dt is None and gemm == "library":
            keep["rs32"], keep["traj32"] = box["rs"], box["traj"]

    leg("config3_rollout_fp32", lambda: rollout_leg("fp32", None, "library"))

    def gae_leg():
        calc_gae = brl_amd.make_calc_gae(cfg32, fp)
        t_gae, (adv, tgt) = timed(lambda: calc_gae(keep["rs32"], keep["traj32"]), 3)
        keep["adv"], keep["tgt"] = adv, tgt
        phases["calc_gae"] = {"ms": t_gae * 1e3, "what": "critic forward on last_obs (fp32) + brl_gae"}

    leg("config3_calc_gae", gae_leg, needs=("config3_rollout_fp32",))

    def update_leg(name, extra):
        cfg = dict(cfg32, **extra)
        net = fp.init(0, device=dev)
        update_step = make_update_step(cfg, fp)
        ubox = {"rs": (net, make_optimizer(cfg, net)) + tuple(keep["rs32"][2:])}

        def do_update():
            ubox["rs"], info = update_step(ubox["rs"], keep["traj32"], keep["adv"], keep["tgt"])
            return info
        t_upd, info = timed(do_update, 3)
        # what the step's launches EXECUTE (r04u_step_pmc.txt: sum of MfmaFlopsF32 = 21.6 GFLOP at minibatch 1024): forward + a weight
        # gradient for every layer + an input gradient for every layer but the first (nobody needs d(loss)/d(obs)); the customary
        # "3 x forward" (22.6 GFLOP) counts a product that is never formed and is kept only as `nominal_3x_forward`
        step_flop = UPDATE_FLOP_PER_SAMPLE * cfg["minibatch_size"]
        uflop = UPDATE_FLOP_PER_SAMPLE * rows * cfg["update_epochs"]
        nominal = 3 * rows * fwd_flop * cfg["update_epochs"]
        graphed = ubox["rs"][1].get("graphed")
        graphed = ubox["rs"][1].get("graphed")
        # config["dw_gemm"] = "bf16x3" (the default): the weight gradients of the four body layers run as SIX bf16 products each on the bf16
        # matrix pipe (brl_mlp_gemm_x3_group); the step's speed of light is then the time of the remaining fp32 products at the
        # v_mfma_f32 peak + six times the weight gradients' FLOPs at the dense bf16 peak
        dw_x3 = bool(getattr(graphed, "dw_x3", False))
        x3_per_sample = 2 * (480 * 1024 + 3 * 1024 * 1024) if dw_x3 else 0
        samples = rows * cfg["update_epochs"]
        t_min = samples * ((UPDATE_FLOP_PER_SAMPLE - x3_per_sample) / 157.3e12 + 6 * x3_per_sample / 2500e12)
        blended = uflop / t_min / 1e12
        phases[name] = {"ms": t_upd * 1e3, "minibatch_steps": nmb, "ms_per_minibatch": t_upd / nmb * 1e3,
                        "gemm_flop_per_step": step_flop, "gemm_tflops": uflop / t_upd / 1e12, "mfma_peak_tflops": blended,
                        "mfma_frac": t_min / t_upd,
                        "fp32_equivalent_vs_f32_peak": uflop / t_upd / 1e12 / 157.3,
                        "roofline": {"bound": "mfma_f32+bf16x3" if dw_x3 else "mfma_f32", "achieved": uflop / t_upd / 1e12, "peak": blended,
                                     "unit": "TFLOP/s", "frac": t_min / t_upd, "flop_per_step": step_flop,
                                     "bf16x3_flop_per_step": x3_per_sample * cfg["minibatch_size"],
                                     "ms_per_step": t_upd / nmb * 1e3,
                                     "what": "fp32-equivalent FLOPs the step's launches form / the step's time; the peak is the one of "
                                             "this mix: the exact-fp32 products at the dense v_mfma_f32 peak (157.3 = 256 CUs x 4 SIMDs "
                                             "x 64 FLOP/clk x 2.4 GHz), the bf16x3 weight gradients as 6 bf16 products at the dense bf16 "
                                             "peak (2500); kernel-by-kernel timeline of the same step: "
                                             "profiles/r06/*_update_timeline.txt"},
                        "nominal_3x_forward": {"gemm_flop_per_step": 3 * fwd_flop * cfg["minibatch_size"],
                                               "gemm_tflops": nominal / t_upd / 1e12, "mfma_frac": nominal / t_upd / 1e12 / 157.3,
                                               "note": "counts layer 0's input gradient, which no launch forms: not a rate the "
                                                       "kernels ran at (rounds 1-4 reported this figure)"},
                        "dtype": "fp32",
                        "path": type(graphed).__name__ if graphed else "eager: " + str(ubox["rs"][1].get("graph_error")),
                        "what": "10 epochs x 256 minibatches of 1024 samples: forward + backward + global-norm clip + Adam"}
        return info

    leg("config3_update", lambda: update_leg("update", {}), needs=("config3_calc_gae",))
    leg("config3_rollout_fp32_bf16x3", lambda: rollout_leg("fp32_bf16x3", None, "bf16x3"))     # (brl_amd's default fp32 rollout)

    # ---- configs[2]: 8192-board duplicate evaluation, two different networks, fp32 (src/evaluation.py:69-204)
    def config2_leg():
        team2 = fp.init(1, device=dev)
        dup = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", NUM_ENVS)
        t_eval, res = timed(lambda: dup(team1, team2, 123), 3)
        out["config2"] = {"workload": "configs[2]: num_envs=8192 duplicate-table evaluation (table A, then seat-swapped table B, "
                                      "IMP), two DeepMind MLPs, greedy, fp32",
                          "ms": t_eval * 1e3, "boards_per_s": NUM_ENVS / t_eval, "dtype": "fp32",
                          "imp_mean": float(res[0][0]), "imp_se": float(res[0][1])}

    leg("config2", config2_leg)

    # ---- the evaluations ppo.py runs EVERY iteration around those three phases (ppo.py:366-381, 461-484; self_play): one
    # simple_evaluate + three simple_duplicate_evaluate (imp_opp, imp_opp_before, imp_opp_after) at num_eval_envs = 10000, fp32
    def evaluators_leg():
        from brl_amd.evaluation import make_evaluate, make_evaluate_log, make_simple_evaluate
        n_eval = int(DEFAULTS.get("num_eval_envs", 10000))
        opp = fp.init(1, device=dev)
        simple = make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", opp, n_eval)
        dup_eval = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", n_eval)
        t_simple, _ = timed(lambda: simple(team1, 7), 3)
        t_dup, _ = timed(lambda: dup_eval(team1, opp, 7), 3)
        # ... and the duplicate evaluation WITH bidding statistics that ppo.py runs every num_eval_step iterations (ppo.py:383-392,
        # src/evaluation.py:207-1032), its 23-entry log_info turned into the eval/... dict on the host: reported, not part of `ms`
        full_eval = make_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", opp, n_eval, duplicate=True)
        t_full, _ = timed(lambda: make_evaluate_log(full_eval(team1, 7)[0]), 3)
        phases["evaluators"] = {"ms": (t_simple + 3 * t_dup) * 1e3, "simple_evaluate_ms": t_simple * 1e3,
                                "simple_duplicate_evaluate_ms": t_dup * 1e3, "duplicate_evaluate_with_statistics_ms": t_full * 1e3,
                                "num_eval_envs": n_eval, "dtype": "fp32",
                                "ms_as_brl_amd_train_plays_them": (t_simple + t_dup) * 1e3,
                                "as_played": "brl_amd.train plays each distinct (parameter version, opponent) pair once: imp_opp of "
                                             "iteration i + 1 IS imp_opp_after of iteration i (same networks, boards, arg-max play), "
                                             "imp_opp_before IS imp_opp unless the pool switched the opponent: 1 + 1 evaluations per "
                                             "iteration in the steady state, 1 + 2 when the opponent changes; `ms` counts the "
                                             "reference's 1 + 3",
                                "what": "per iteration (ppo.py:366-381,461-484, self_play): jit_simple_evaluate + 3 x "
                                        "jit_simple_duplicate_evaluate; the full duplicate evaluation with statistics runs every "
                                        "num_eval_step iterations only and is not included"}

    leg("config3_evaluators", evaluators_leg)

    # ---- configs[4] rehearsal on ONE GPU: the compute side of a rank's minibatch step under a process group (world = 8
    # geometry: buckets, slices, the norm's partials, grad_scale), every collective replaced by a no-op with the same stream
    # ordering inside the step's graph
    def rehearsal_leg():
        out["config4_rehearsal"] = rehearse_multirank(torch, dev, cfg32, fp, keep["traj32"], keep["adv"], keep["tgt"],
                                                      phases["update"]["ms_per_minibatch"])

    leg("config4_rehearsal", rehearsal_leg, needs=("config3_update",))

    # ---- the same iteration with ppo.py's other architecture, actor_model_type = "FAIR" (src/models.py:34-69): rollout through
    # brl_fair_forward (one launch per forward), update through FusedFair (brl_fair_chain + brl_mlp_gemm_group: five launches per step)
    def fair_leg():
        fpf = make_forward_pass("relu", "FAIR")
        netf = fpf.init(0, device=dev)
        roll_f = brl_amd.make_roll_out(cfg32, env, fpf, fpf)
        stf = env.init(0, num_envs=NUM_ENVS)
        fbox = {"rs": (netf, None, stf, stf.observation, 0, 0)}

        def do_roll_f():
            fbox["rs"], fbox["traj"] = roll_f(fbox["rs"], netf)
            return None
        t_roll_f, _ = timed(do_roll_f, 3)
        advf, tgtf = brl_amd.make_calc_gae(cfg32, fpf)(fbox["rs"], fbox["traj"])
        upd_f = make_update_step(cfg32, fpf)
        fu = {"rs": (netf, make_optimizer(cfg32, netf)) + tuple(fbox["rs"][2:])}

        def do_update_f():
            fu["rs"], info = upd_f(fu["rs"], fbox["traj"], advf, tgtf)
            return info
        t_upd_f, _ = timed(do_update_f, 3)
        gf = fu["rs"][1].get("graphed")
        out["config3_fair"] = {"workload": "configs[3]'s iteration with actor_model_type = FAIR (eleven 200-wide layers, residual; "
                                           "0.60 M parameters), fp32",
                               "rollout_ms": t_roll_f * 1e3, "update_ms": t_upd_f * 1e3, "ms_per_minibatch": t_upd_f / nmb * 1e3,
                               "iteration_ms": (t_roll_f + t_upd_f) * 1e3 + phases.get("calc_gae", {}).get("ms", 0.0),
                               "path": type(gf).__name__ if gf else "eager: " + str(fu["rs"][1].get("graph_error")),
                               "what": "rollout: brl_fair_forward per forward; update: brl_fair_chain (forward + loss + backward chain, "
                                       "16 rows per workgroup) + brl_mlp_gemm_group (12 weight gradients) + finalize + clip / Adam"}

    leg("config3_fair", fair_leg)
    leg("config3_rollout_bf16", lambda: rollout_leg("bf16", "bf16"))

    c3 = dict(phases, workload="configs[3]: ppo.py iteration at num_envs=8192, num_steps=32, minibatch 1024, "
                               "10 epochs, DeepMind MLP")
    if all(k in phases for k in ("rollout_fp32", "calc_gae", "update")):
        it32 = phases["rollout_fp32"]["ms"] + phases["calc_gae"]["ms"] + phases["update"]["ms"]
        c3.update(iteration_ms_fp32=it32, iteration_macro_steps_per_s_fp32=rows / (it32 * 1e-3))
        if "evaluators" in phases:
            full32 = it32 + phases["evaluators"]["ms"]
            c3.update(evaluators_ms=phases["evaluators"]["ms"], iteration_ms_fp32_full=full32,
                      iteration_macro_steps_per_s_fp32_full=rows / (full32 * 1e-3))
        if "rollout_fp32_bf16x3" in phases:
            itx = phases["rollout_fp32_bf16x3"]["ms"] + phases["calc_gae"]["ms"] + phases["update"]["ms"]
            c3.update(iteration_ms_fp32_bf16x3_rollout=itx, iteration_macro_steps_per_s_fp32_bf16x3_rollout=rows / (itx * 1e-3))
        if "rollout_bf16" in phases:
            it16 = phases["rollout_bf16"]["ms"] + phases["calc_gae"]["ms"] + phases["update"]["ms"]
            c3.update(iteration_ms_bf16_rollout=it16, iteration_macro_steps_per_s_bf16_rollout=rows / (it16 * 1e-3))
    out["config3"] = c3
    out["legs_s"] = legs_s
    out["dropped"] = dropped
    if failed:
        out["failed"] = failed
    out["secondary_s"] = round(time.perf_counter() - t_begin, 1)
    return out


def rehearse_multirank(torch, dev, cfg, fp, traj, adv, tgt, single_ms):
    """ms per minibatch step of FusedMinibatch as ONE RANK OF EIGHT runs it (configs[4]'s geometry: five buckets x eight slices),
    measured on one GPU: both forms of config["grad_allreduce"], 256 steps (one epoch of 8192 x 32 at minibatch 1024), every
    collective a no-op that keeps the stream edges of an asynchronous RCCL collective and sits INSIDE the eight-step hipGraph like
    the real ones (RCCL's collectives record into a graph: profiles/r05/r05a_rccl_capture_probe.txt).  What crosses xGMI is NOT in
    this number — DESIGN §7 adds it; the parameters it leaves are not a real update (nothing is reduced or gathered)."""
    from brl_amd.roll_out import Transition
    from brl_amd.update import FusedMinibatch, make_optimizer
    rows = NUM_ENVS * NUM_STEPS
    flat = Transition(*[x.reshape((rows,) + x.shape[2:]) for x in traj])
    adv_f, tgt_f = adv.reshape(rows), tgt.reshape(rows)
    side = torch.cuda.Stream()
    world, rank = 8, 3

    class _Work:                       # like a collective's Work: waits for ITS collective's end, not for the whole stream
        def __init__(self):
            self.done = torch.cuda.Event()
            self.done.record(side)

        def wait(self):
            torch.cuda.current_stream().wait_event(self.done)

    class NoopCollectives:
        capturable = True

        def __init__(self):
            self.rank, self.world, self.calls = rank, world, 0

        def _edge(self, async_op):
            self.calls += 1
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)          # the collective starts behind the kernels that produced its bucket
            if async_op:
                return _Work()
            cur.wait_stream(side)          # a blocking collective: what follows waits for it
            return None

        def all_reduce(self, t, async_op):
            return self._edge(async_op)

        def reduce_scatter(self, out, inp, async_op):
            return self._edge(async_op)

        def all_gather(self, out, inp, async_op):
            return self._edge(async_op)

    res = {"what": "one rank's minibatch step with world = 8 geometry on one GPU, collectives = no-ops with an asynchronous "
                   "collective's stream edges, captured inside the eight-step hipGraph; 256 steps, median of 3; ring traffic per "
                   "step and rank at world 8: 2 * 7/8 * 14.7 MB = 25.8 MB in both forms ('flat' is the default; what a collective's "
                   "kernel costs the GEMMs beside it is measured by scripts/overlap_probe.py: profiles/r05/r05i_overlap_probe.txt)",
           "single_rank_ms_per_minibatch": single_ms}
    for mode in ("flat", "sharded"):
        net = fp.init(0, device=dev)
        opt = make_optimizer(cfg, net)["opt"]
        co = NoopCollectives()
        fm = FusedMinibatch(dict(cfg, grad_allreduce=mode), net, opt, 1024, dev, world=world, collective=co)
        per_step = sum(1 for item in fm.program if item[0] == "c")
        ts = []
        for rep in range(4):
            perms = [torch.randperm(rows, device=dev)]
            fm.begin_update(flat, adv_f, tgt_f, perms)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fm.run_steps(rows // 1024)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            fm.end_update()
        t = float(np.median(ts[1:]))
        res[mode] = {"ms_per_minibatch": t / (rows // 1024) * 1e3, "graph_replays_per_step": 1.0 / fm.graph_steps if fm.in_graph else None,
                     "collectives_per_step": per_step, "collectives_inside_the_graph": bool(fm.in_graph),
                     "adam_sweep_fraction": 1.0 / world if mode == "sharded" else 1.0,
                     "overhead_vs_single_rank_ms": t / (rows // 1024) * 1e3 - single_ms}
        del fm, net, opt
    return res


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
        if os.environ.get("BRL_TUNABLEOP_FILE"):
            torch.cuda.tunable.set_filename(os.environ["BRL_TUNABLEOP_FILE"])
    # BRL_BENCH_PPO_ENVS / _EPOCHS: a REDUCED size for rehearsals of the N-rank path on a box with fewer GPUs than ranks (gloo
    # ranks sharing a device: tests/test_multi_gpu_rccl.py); the record then says so and is not a measurement of configs[3] / [4]
    n_envs = int(os.environ.get("BRL_BENCH_PPO_ENVS", NUM_ENVS))
    epochs = int(os.environ.get("BRL_BENCH_PPO_EPOCHS", 10))
    # BRL_GRAD_ALLREDUCE=flat|sharded: the form of the multi-rank gradient step (default: brl_amd's, "flat")
    if os.environ.get("BRL_GRAD_ALLREDUCE"):
        DEFAULTS = dict(DEFAULTS, grad_allreduce=os.environ["BRL_GRAD_ALLREDUCE"])
    cfg = dict(DEFAULTS, num_envs=n_envs, num_steps=NUM_STEPS, minibatch_size=1024, update_epochs=epochs,
               inference_dtype=(os.environ.get("BRL_INFER_DTYPE", "fp32").replace("fp32", "") or None), graph_rollout=True)
    # (default = the reference's fp32 forwards; BRL_INFER_DTYPE=bf16 / fp16 = the opt-in, narrower inference path)
    cfg["num_minibatches"] = cfg["num_envs"] * cfg["num_steps"] // cfg["minibatch_size"]
    keys, values = synthetic_lut(LUT_LEN, 0)
    env = brl_amd.BridgeBidding(lut=(keys, values), device=dev, env_offset=rank * n_envs)
    fp = make_forward_pass("relu", "DeepMind")
    params = fp.init(0, device=dev)
    opt_state = make_optimizer(cfg, params)
    roll_out = brl_amd.make_roll_out(cfg, env, fp, fp)
    calc_gae = brl_amd.make_calc_gae(cfg, fp)
    update_step = make_update_step(cfg, fp)
    st = env.init(0, num_envs=n_envs)
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
    rows = n_envs * NUM_STEPS
    fwd_flop = 2 * 3_677_184                      # 7.354 MFLOP per forward per sample
    roll_flop = 4 * rows * fwd_flop + n_envs * fwd_flop
    upd_flop = UPDATE_FLOP_PER_SAMPLE * rows * cfg["update_epochs"]   # executed: no input gradient for layer 0 (see bench_secondary)
    return {
        "metric": "ppo.py iteration macro-steps/sec at num_envs=8192, num_steps=32, minibatch 1024, 10 epochs (secondary, configs[3])",
        "value": world * rows * iters / elapsed, "unit": "macro-steps/s", "n_gpus": world, "steps": iters, "warmup": 1,
        "ms_per_step": elapsed / iters * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"rollout inference {cfg['inference_dtype'] or 'fp32 (hidden layers: ' + str(cfg.get('inference_gemm') or 'bf16x3') + ')'}, update fp32", "data": "synthetic",
        "config": {"workload": "configs[3]: roll_out (policy in the loop, competitive) + calc_gae + update_step",
                   "num_envs_per_gpu": n_envs, "num_steps": NUM_STEPS, "minibatch_size": 1024, "update_epochs": epochs,
                   "graph_rollout": True, "grad_allreduce": getattr(rs[1].get("graphed"), "allreduce_mode", None),
                   "collectives_inside_the_graph": getattr(rs[1].get("graphed"), "in_graph", None),
                   "rehearsal_size": (n_envs, epochs) != (NUM_ENVS, 10)},
        "phases_ms": {k: v * 1e3 for k, v in med.items()},
        "rollout": {"macro_steps_per_s": rows / med["rollout"], "raw_env_steps_per_s": 4 * rows / med["rollout"],
                    "gemm_tflops": roll_flop / med["rollout"] / 1e12,
                    "mfma_peak_tflops": 2500.0 if cfg["inference_dtype"] in ("bf16", "fp16") else 157.3},
        "update": {"gemm_tflops": upd_flop / med["update"] / 1e12, "mfma_peak_tflops": 157.3,
                   "gemm_flop_per_step": UPDATE_FLOP_PER_SAMPLE * 1024,
                   "note": "fp32 GEMMs as executed (forward, dW for every layer, dX for every layer but the first), "
                           f"{epochs * rows // 1024} minibatch steps of 1024 samples, hipGraph-replayed"},
    }


if __name__ == "__main__":
    main()
