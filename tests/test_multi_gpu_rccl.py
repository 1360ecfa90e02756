"""-m gpu: the RCCL path on real links — `bench.py --gpus 2` end to end over "nccl", one GPU per rank (BASELINE.json
configs[4] is the same launch at N = 8).  SKIPPED with a reason on a box with fewer than 2 GPUs; the N-rank control flow
is covered there by the gloo rehearsal below and by tests/test_bench_launcher.py on CPU.  (The two-rank ppo.py loop and
the gradient collectives over RCCL: tests/test_gpu_parity.py::test_ppo_loop_two_ranks_rccl,
::test_fused_update_with_gradient_collectives_two_ranks_rccl.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
DROP = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def _bench(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in DROP}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_over_rccl_lists_two_distinct_devices():
    n = torch.cuda.device_count()   # (does not initialise the GPU)
    if n < 2:
        pytest.skip(f"RCCL path needs >= 2 GPUs (one per rank); this box has {n}")
    r, out = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"])
    assert r.returncode == 0 and out is not None, r.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["ranks_backend"].startswith("RCCL") and "ranks_error" not in out
    ids = [x["pci_bus_id"] or x["uuid"] for x in out["ranks"]]
    assert len(out["ranks"]) == 2 and len(set(ids)) == 2 and None not in ids, out["ranks"]
    assert [x["device_index"] for x in out["ranks"]] == [0, 1]
    assert out["config"]["env_offsets"] == [0, 8192]
    one = 8192 * 32 / (max(x["local_ms_per_step"] for x in out["ranks"]) * 1e-3)
    assert 1.5 * one < out["value"] <= 2.0 * one * 1.001       # whole-job value = both shards over the slowest rank's time


def test_bench_gpus_2_gloo_rehearsal_on_one_gpu():
    """The same N-rank control flow on THIS box's single GPU (gloo control plane, ranks sharing the device): the launcher,
    per-rank shards, barrier + max-over-ranks, the `ranks` records — labelled as a rehearsal, never as an RCCL run."""
    r, out = _bench(["--gpus", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"],
                    BRL_BENCH_BACKEND="gloo")
    assert r.returncode == 0 and out is not None, r.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["ranks_backend"].startswith("gloo") and len(out["ranks"]) == 2
    assert all(x["kernel_ms"] and x["kernel_ms"] > 0 for x in out["ranks"][:1])
    assert out["roofline"]["frac"] > 0.05


def test_rccl_run_with_one_device_for_two_ranks_is_refused(monkeypatch):
    """Rank proof: if the records do not name N distinct GPUs under RCCL the bench says so and exits non-zero — checked on
    the record-checking function itself (no second GPU needed)."""
    sys.path.insert(0, ROOT)
    import bench

    class _Dist:
        @staticmethod
        def all_gather_object(recs, rec):
            recs[0], recs[1] = dict(rec, rank=0), dict(rec, rank=1)     # both ranks report the same device

    recs, err = bench.gather_ranks(torch, _Dist, 0, 2, torch.device("cuda", 0), 1.0, 0.025, 20, "nccl")
    assert err is not None and "distinct" in err and len(recs) == 2
    recs, err = bench.gather_ranks(torch, _Dist, 0, 2, torch.device("cuda", 0), 1.0, 0.025, 20, "gloo")
    assert err is None


def test_bench_ppo_four_gloo_ranks_on_one_gpu():
    """configs[4]'s measurement path — `bench.py --gpus N --config ppo`: per-rank env shards (env_offset = rank * num_envs), the
    policy rollout, calc_gae, the fused update with its gradient collectives (FusedMinibatch(world = N), here the sharded form:
    BRL_GRAD_ALLREDUCE; the default "flat" form runs in the two-rank loop), barrier + max-over-ranks, ONE JSON line — rehearsed with FOUR gloo ranks sharing this box's GPU at a
    reduced size (1024 tables, 2 epochs).  Four, not eight: a GPU box of this pool allows at most 6 processes on its card; the
    world = 8 geometry itself runs in one process (test_fused_update_sharded_geometry_of_eight_ranks_on_one_gpu,
    bench.py's config4_rehearsal).  Not a measurement: the record says so (`rehearsal_size`)."""
    r, out = _bench(["--gpus", "4", "--config", "ppo", "--steps", "1"], BRL_BENCH_BACKEND="gloo", BRL_BENCH_PPO_ENVS="1024",
                    BRL_BENCH_PPO_EPOCHS="2", BRL_GRAD_ALLREDUCE="sharded")
    assert r.returncode == 0 and out is not None, r.stderr[-3000:]
    assert out["n_gpus"] == 4 and out["config"]["rehearsal_size"] is True and out["config"]["num_envs_per_gpu"] == 1024
    assert out["config"]["grad_allreduce"] == "sharded" and out["config"]["collectives_inside_the_graph"] is False   # gloo: eager
    assert out["value"] > 0 and all(v > 0 for v in out["phases_ms"].values())


def test_ppo_loop_with_every_collective_over_rccl_at_world_1():
    """brl_amd.train with BRL_FORCE_DIST=1 under a world-1 RCCL process group (scripts/soak_train_rccl_world1.py, 4 iterations x 1024
    tables, both forms of the gradient step): parameter broadcast, barriers, the sharded evaluators' all-reduces, the opponent-index
    broadcast, rollout captures, the update's graph with RCCL nodes inside, rank-sync checksums, the optimizer-state gather — the
    ORDER of eager collectives, captures and replays under ProcessGroupNCCL's watchdog thread, which is what a node run shares with
    this box (an eager collective polled by the watchdog during a capture that contains collectives aborts the process: the guard in
    FusedStep, profiles/r05/r05n_rccl_eager_then_capture_probe.txt)."""
    env = {k: v for k, v in os.environ.items() if k not in DROP}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_train_rccl_world1.py"), "4", "1024"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if "iterations under a world-1 RCCL process group" in l]
    assert len(lines) == 2 and all("collectives inside the graph: True" in l and "backend nccl" in l for l in lines), r.stdout[-2000:]
    assert "mode flat" in lines[0] and "mode sharded" in lines[1]


def test_capture_group_needs_no_sleep_between_an_eager_collective_and_a_capture(tmp_path):
    """Round 5 fenced FusedStep's captures with synchronize + 0.3 s of sleep (an eager collective still in ProcessGroupNCCL's watchdog
    list when RCCL's stream starts to capture aborts the process).  Round 6: the captured collectives have a process group of their
    own (fused_update._capture_group) that never carries an eager one.  scripts/rccl_eager_then_capture_probe.py case I: twenty times
    in one process an eager all-reduce on the default group, AT ONCE a 0.6 s capture with a collective inside — no synchronize, no
    sleep; case J the same behind a throw-away capture (FusedStep's phase 2); case F (one group, no fence) is the control that
    aborts where the runtime still behaves as in round 5."""
    env = {k: v for k, v in os.environ.items() if k not in DROP}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", CASES="IJ", REPEAT="20")
    out = tmp_path / "probe.txt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_eager_then_capture_probe.py"), str(out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    text = out.read_text()
    assert "case I: survived" in text and "case J: survived" in text, text


def test_node_day_script_dry_run(tmp_path):
    """scripts/node_day.sh — the one command for the day an 8-GPU node exists (RCCL tests un-skipped, bench.py --gpus 1,2,4,8,
    bench.py --config ppo in both forms of the gradient step, the 14.7 MB all-reduce inside a graph) — executed end to end at world 1
    (NODE_DRY=1: BRL_FORCE_DIST=1, reduced ppo sizes): the script itself has run, every output file parses."""
    env = {k: v for k, v in os.environ.items() if k not in DROP}
    env.update(NODE_DRY="1", PYTHON=sys.executable)
    out = tmp_path / "node"
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "node_day.sh"), str(out)], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=1500)
    summary = (out / "summary.txt").read_text() if (out / "summary.txt").exists() else ""
    assert r.returncode == 0 and "node_day: done" in summary, (r.stdout[-2000:], r.stderr[-2000:])
    b = json.loads((out / "bench_gpus1.json").read_text())
    assert b["n_gpus"] == 1 and b["value"] > 0 and b["ranks_backend"].startswith("RCCL")
    for mode in ("flat", "sharded"):
        d = json.loads((out / f"bench_ppo_{mode}_gpus1.json").read_text())
        assert d["config"]["grad_allreduce"] == mode and d["config"]["collectives_inside_the_graph"] is True and d["value"] > 0
    p = json.loads((out / "allreduce_graph_probe.json").read_text())
    assert p["world"] == 1 and p["us_per_collective"]["all_reduce_flat_14.7MB"] > 0
    assert " passed" in (out / "pytest_rccl.txt").read_text().splitlines()[-1]


def test_peer_pointer_optimizer_prototype_two_processes_one_gpu(tmp_path):
    """scripts/micro/peer_adam.hip ("what comes next": the gradient collective fused into the optimizer launches — every
    rank maps its peers' gradient / parameter buffers with hipIpcOpenMemHandle, reduces ITS slice straight from them, runs clip + Adam on
    it and stores the new parameters into every peer; three monotone flag words per peer instead of collective kernels and
    cross-stream edges).  A prototype outside the library, never a default: this is its functional proof — two processes on the one
    GPU, three steps: no wait times out, both ranks end with bit-identical parameters equal to a float64 restatement, the moments
    live on the owner's slice only."""
    env = {k: v for k, v in os.environ.items() if k not in DROP}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = tmp_path / "peer.txt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "peer_adam_probe.py"), "2", str(1 << 20), str(out)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and out.read_text().strip().endswith("PASS"), (r.stdout[-2000:], r.stderr[-2000:])


def test_bench_control_flow_over_rccl_at_world_1():
    """The N-rank control flow of `bench.py` (process-group set-up with the device bound, barriers, MAX over ranks of the step time,
    the per-rank records gathered with all_gather_object) really over RCCL with one peer (BRL_FORCE_DIST=1): what the driver's N > 1
    runs execute besides the step itself."""
    r, out = _bench(["--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"], BRL_FORCE_DIST="1")
    assert r.returncode == 0 and out is not None, r.stderr[-3000:]
    assert out["n_gpus"] == 1 and out["ranks_backend"].startswith("RCCL") and len(out["ranks"]) == 1 and "ranks_error" not in out
    assert out["value"] > 0 and out["roofline"]["frac"] > 0.05


def test_bench_ppo_over_rccl_at_world_1():
    """`bench.py --config ppo` (configs[4]'s measurement path) with its gradient all-reduce a REAL RCCL node of the update's hipGraph
    (world 1, BRL_FORCE_DIST=1), at a reduced size: the record names the form and says the collectives are inside the graph."""
    r, out = _bench(["--gpus", "1", "--config", "ppo", "--steps", "1"], BRL_FORCE_DIST="1", BRL_BENCH_PPO_ENVS="1024", BRL_BENCH_PPO_EPOCHS="2")
    assert r.returncode == 0 and out is not None, r.stderr[-3000:]
    assert out["config"]["grad_allreduce"] == "flat" and out["config"]["collectives_inside_the_graph"] is True
    assert out["value"] > 0 and out["config"]["rehearsal_size"] is True
