"""The one number DESIGN §7's multi-GPU estimate rests on: what the PPO step's gradient collectives cost INSIDE a hipGraph over xGMI.
Every rank (one per GPU, RCCL):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                                     --master-port P scripts/allreduce_graph_probe.py [out.json]
(or plain `python scripts/allreduce_graph_probe.py` = world 1: the dry run scripts/node_day.sh executes on a one-GPU box.)
Measured, per collective, MAX over ranks: the 14.7 MB flat gradient of the DeepMind MLP (3 676 199 floats, brl_amd.fused_update's
"flat" form) as ONE all-reduce; the "sharded" form's buckets — reduce-scatter + all-gather of 4.2 MB (a hidden layer) and 2.1 MB (the
tail) — each captured 8 times in one graph (as the step's eight-step graph carries them) on the capture-only process group
(fused_update._capture_group), replayed 20 times; and the same all-reduce issued eagerly, for scale."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from brl_amd.fused_update import _capture_group
    cg = _capture_group()
    n_flat = 3 * 1024 * 1024 + 480 * 1024 + 39 * 1024 + 4 * 1024 + 39          # W_1..3, W_0, heads, biases
    n_flat = (n_flat + 4 * world - 1) // (4 * world) * (4 * world)
    sizes = {"all_reduce_flat_14.7MB": n_flat, "bucket_hidden_4.2MB": 1024 * 1024, "bucket_tail_2.1MB": (n_flat - 3 * 1024 * 1024)}
    K, REPLAYS = 8, 20
    out = {"world": world, "backend": "nccl (RCCL)", "graph_copies": K, "replays": REPLAYS, "us_per_collective": {}}

    def max_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_graph(body):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):        # (the first capture of a collective on the group is also RCCL's set-up for its size)
            body()
        g.replay()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(K):
                body()
        g.replay()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPLAYS):
            g.replay()
        torch.cuda.synchronize()
        return max_ranks((time.perf_counter() - t0) / (REPLAYS * K) * 1e6)

    for name, n in sizes.items():
        n = n // (4 * world) * (4 * world)
        buf = torch.ones(n, device=dev)
        mine = buf[rank * (n // world):(rank + 1) * (n // world)]
        if name.startswith("all_reduce"):
            out["us_per_collective"][name] = timed_graph(lambda: dist.all_reduce(buf, group=cg))
            torch.cuda.synchronize()
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            out["us_per_collective"][name + "_eager_default_group"] = max_ranks((time.perf_counter() - t0) / 20 * 1e6)
        else:
            out["us_per_collective"][name + "_reduce_scatter"] = timed_graph(lambda: dist.reduce_scatter_tensor(mine, buf, group=cg))
            out["us_per_collective"][name + "_all_gather"] = timed_graph(lambda: dist.all_gather_into_tensor(buf, mine, group=cg))
        out.setdefault("bytes", {})[name] = 4 * n
    ar = out["us_per_collective"]["all_reduce_flat_14.7MB"]
    out["all_reduce_flat_algbw_GBps"] = 4 * sizes["all_reduce_flat_14.7MB"] / (ar * 1e-6) / 1e9
    out["all_reduce_flat_busbw_GBps"] = out["all_reduce_flat_algbw_GBps"] * 2 * (world - 1) / world
    if rank == 0:
        s = json.dumps(out)
        print(s, flush=True)
        if len(sys.argv) > 1:
            open(sys.argv[1], "w").write(s + "\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
