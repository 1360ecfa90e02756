"""Multi-GPU plumbing of the hot path: tables are independent, so the path shards
with NO data-path collective — rank r owns global tables [r*num_envs, (r+1)*num_envs) and draws
its boards/actions from the same counter-based streams a single process would use for them.
The only collectives are bookkeeping: max-over-ranks wall time, summed counters."""
from __future__ import annotations

import os


def distributed() -> bool:
    """A process group exists AND its collectives are to be issued: world_size > 1 — or BRL_FORCE_DIST=1, the world-1 REHEARSAL of
    the multi-rank control flow (every collective of the loop really goes through RCCL with one peer: scripts/soak_train_rccl_world1.py;
    how the eager-collective-before-capture abort was found, profiles/r05/r05_experiments.txt section 10)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("BRL_FORCE_DIST") == "1")


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_offset(rank: int, num_envs_per_rank: int) -> int:
    """env_offset of rank `rank`: global index of its table 0."""
    return rank * num_envs_per_rank


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_int(value: int, device=None, src: int = 0) -> int:
    """rank `src`'s integer on every rank (e.g. the index of the FSP / PFSP opponent rank 0 drew, ppo.py:376-460)."""
    import torch
    import torch.distributed as dist

    if not distributed():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    return int(t.item())


def broadcast_parameters(module, src: int = 0) -> None:
    """rank `src`'s weights on every rank, once at start: the ranks build the same network from the same seed
    or file already; this makes the invariant explicit before the first all-reduced update."""
    import torch
    import torch.distributed as dist

    if not distributed():
        return
    with torch.no_grad():
        for p in module.parameters():
            dist.broadcast(p.data, src=src)
