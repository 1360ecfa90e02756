"""world_size-2 gloo tests of the N>1 path (CPU): env shards are independent — two ranks with
env_offset = rank * n reproduce exactly what one process computes for 2n tables — and the
bench's cross-rank reductions behave.  The per-shard compute here is the oracle (no GPU in this
container); the sharding contract (env_offset semantics) is the same one libbrl_hip implements and
tests/test_gpu_parity.py::test_init_random_env_offset checks on the device."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PER_RANK, T, SEED = 96, 12, 4242


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brl_amd.dist import max_over_ranks, rank_world, shard_offset, sum_over_ranks
    from oracle import Oracle

    assert rank_world() == (rank, world)
    d = np.load(os.path.join(ROOT, "tests", "golden", "wb5_dds_1000.npz"))
    orc = Oracle(d["keys"], d["values"])
    off = shard_offset(rank, N_PER_RANK)
    st = orc.init_random(N_PER_RANK, seed=SEED, env_offset=off)
    out = orc.rollout_random(st, T, seed=SEED, env_offset=off)
    # no data-path collective: only the bookkeeping reductions of bench.py
    t = max_over_ranks(1.0 + rank)
    total = sum_over_ranks(float(out["terminated_count"]))
    # gather shards for the comparison with the single-process run
    obs = torch.from_numpy(out["obs"])
    gathered = [torch.empty_like(obs) for _ in range(world)]
    dist.all_gather(gathered, obs)
    if rank == 0:
        np.save(os.path.join(out_dir, "obs.npy"), torch.cat(gathered, dim=1).numpy())
        np.save(os.path.join(out_dir, "meta.npy"), np.array([t, total]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_one_process(tmp_path, oracle):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    obs = np.load(tmp_path / "obs.npy")
    t_max, tc_sum = np.load(tmp_path / "meta.npy")
    st = oracle.init_random(world * N_PER_RANK, seed=SEED)
    want = oracle.rollout_random(st, T, seed=SEED)
    assert obs.shape == want["obs"].shape and np.array_equal(obs, want["obs"])
    assert t_max == 2.0 and tc_sum == want["terminated_count"]
