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


# ---------------------------------------------------------------------------------------------------------------
# the gradient step's collective choreography (brl_amd/fused_update.py: "flat" and "sharded") over REAL gloo collectives, with
# the CPU shim's clip + Adam on rank slices (oracle/brl_shim.c: brl_adam_shard_norm / _apply) in place of the device kernels
# ---------------------------------------------------------------------------------------------------------------
def _grad_step_worker(rank, world, port, out_dir):
    import ctypes
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from brl_amd._capi import ShardGeom
    from oracle.binding import shim_path
    oracle.build()
    shim = ctypes.CDLL(shim_path())
    f32, vp, i32, i64 = ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    shim.brl_adam_shard_norm.argtypes = [i32, vp, ctypes.POINTER(ShardGeom), i32, i32, f32, vp, vp, vp, vp]
    shim.brl_adam_shard_apply.argtypes = [i32, vp, vp, vp, vp, ctypes.POINTER(ShardGeom), i32, i32, vp, vp, f32, vp, f32, f32, f32, f32, f32, vp,
                                          vp, i64, vp]
    lens = [48, 16, 8]                               # three buckets: slices of 48 / 16 / 8 floats per rank
    geom = ShardGeom()
    geom.nbuckets, geom.world, geom.nsub = len(lens), world, 2
    offs, off = [], 0
    for b, ln in enumerate(lens):
        geom.off[b], geom.len[b] = off, ln
        offs.append(off)
        off += world * ln
    n = off
    npart = world * len(lens) * 2
    rng = np.random.default_rng(0)
    p0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    res = {}
    for mode in ("flat", "sharded"):
        p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
        step, norm = torch.zeros(1), torch.zeros(1)
        grng = np.random.default_rng(100 + rank)     # every rank its own gradients, the same in both modes
        for it in range(3):
            g = torch.from_numpy((grng.standard_normal(n) * (1e-3 if it == 1 else 1.0)).astype(np.float32))
            part = torch.full((npart,), float("nan"))
            if mode == "flat":
                dist.all_reduce(g)                                                              # SUM: the sweep scales by 1 / world
                assert shim.brl_adam_shard_norm(0, ptr(g), ctypes.byref(geom), 0, world, f32(1.0 / world), ptr(part), ptr(step), None, None) == 0
                lo, hi = 0, world
            else:
                for b, ln in enumerate(lens):                                                   # reduce-scatter per bucket, in place
                    bucket = g[offs[b]:offs[b] + world * ln]
                    dist.reduce_scatter_tensor(bucket[rank * ln:(rank + 1) * ln], bucket)
                assert shim.brl_adam_shard_norm(0, ptr(g), ctypes.byref(geom), rank, rank + 1, f32(1.0 / world), ptr(part), ptr(step), None, None) == 0
                per = npart // world
                dist.all_gather_into_tensor(part, part[rank * per:(rank + 1) * per])    # the partials of the other ranks' slices
                lo, hi = rank, rank + 1
            assert not torch.isnan(part).any()
            assert shim.brl_adam_shard_apply(0, ptr(p), ptr(g), ptr(m), ptr(v), ctypes.byref(geom), lo, hi, ptr(part), ptr(step), f32(1e-3), None,
                                             f32(0.9), f32(0.999), f32(1e-5), f32(0.5), f32(1.0 / world), ptr(norm), None, 0, None) == 0
            if mode == "sharded":
                for b, ln in enumerate(lens):                                                   # parameters back, bucket by bucket, in place
                    bucket = p[offs[b]:offs[b] + world * ln]
                    dist.all_gather_into_tensor(bucket, bucket[rank * ln:(rank + 1) * ln])
        if mode == "sharded":                                                                   # gather_optimizer_state
            for t_ in (m, v):
                for b, ln in enumerate(lens):
                    bucket = t_[offs[b]:offs[b] + world * ln]
                    dist.all_gather_into_tensor(bucket, bucket[rank * ln:(rank + 1) * ln])
        res[mode] = (p.clone(), m.clone(), v.clone(), float(norm))
    torch.save(res, os.path.join(out_dir, f"step{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_step_choreography_flat_equals_sharded_over_gloo(tmp_path):
    """Two gloo ranks with DIFFERENT gradients run three clip + Adam steps in both forms of the multi-rank step — all-reduce +
    replicated sweep, and reduce-scatter per bucket + the rank's slices + all-gather of partials and parameters (in place, as
    brl_amd/fused_update.py issues them): bit-identical parameters and moments in both forms, on both ranks, and equal to
    torch.optim.Adam + clip_grad_norm_ on the mean gradient."""
    world = 2
    mp.start_processes(_grad_step_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"step{k}.pt") for k in range(world)]
    for mode in ("flat", "sharded"):
        for a, b in zip(r[0][mode][:3], r[1][mode][:3]):
            assert torch.equal(a, b), mode
    for a, b in zip(r[0]["flat"][:3], r[0]["sharded"][:3]):
        assert torch.equal(a, b)
    assert r[0]["flat"][3] == r[0]["sharded"][3]
    n = r[0]["flat"][0].numel()
    rng = np.random.default_rng(0)
    ref = torch.nn.Parameter(torch.from_numpy(rng.standard_normal(n).astype(np.float32)))
    opt = torch.optim.Adam([ref], lr=1e-3, eps=1e-5)
    grngs = [np.random.default_rng(100 + k) for k in range(world)]
    for it in range(3):
        gs = [torch.from_numpy((g.standard_normal(n) * (1e-3 if it == 1 else 1.0)).astype(np.float32)) for g in grngs]
        ref.grad = (gs[0] + gs[1]) / world
        want_norm = float(torch.nn.utils.clip_grad_norm_([ref], 0.5))
        opt.step()
    assert abs(r[0]["flat"][3] - want_norm) < 1e-5 * want_norm
    assert torch.allclose(r[0]["flat"][0], ref.detach(), atol=2e-6)
