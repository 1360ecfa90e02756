"""Same-process A/B of the PPO minibatch step (brl_amd.update.FusedMinibatch) under different configs, interleaved rounds:
    python scripts/step_ab.py own_gemm=0 own_gemm=1 [rounds=5] [steps=256]
Each variant is `key=value[,key=value...]` over the configs[3] settings (8192 x 32 synthetic trajectory, minibatch 1024, DeepMind
MLP, fp32); prints ms per minibatch step (median / min over the rounds) and, with `check=1`, the max parameter difference between
the variants after the same steps from the same start (should be rounding-level).  `rccl=1`: under a world-1 RCCL process group
(variants with `force_collectives=1,grad_allreduce=sharded|flat` then carry real collective nodes in their graphs)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brl_amd.models import make_forward_pass   # noqa: E402
from brl_amd.roll_out import Transition        # noqa: E402
from brl_amd.train import DEFAULTS             # noqa: E402
from brl_amd.update import FusedMinibatch, make_optimizer   # noqa: E402


def parse(v):
    d = {}
    for kv in v.split(","):
        k, x = kv.split("=")
        d[k] = {"0": False, "1": True}.get(x, x)
        if isinstance(d[k], str):
            try:
                d[k] = int(x)
            except ValueError:
                try:
                    d[k] = float(x)
                except ValueError:
                    pass
    return d


def main():
    args = [a for a in sys.argv[1:]]
    opts = {"rounds": 5, "steps": 256, "check": 0, "rccl": 0}
    variants = []
    for a in args:
        k = a.split("=")[0]
        if k in opts and "," not in a:
            opts[k] = int(a.split("=")[1])
        else:
            variants.append(a)
    dev = torch.device("cuda", 0)
    if opts["rccl"]:   # a world-1 RCCL process group: `force_collectives=1,grad_allreduce=sharded` then runs the multi-rank program with
        import torch.distributed as dist   # real (single-peer) reduce-scatter / all-gather nodes inside the step's hipGraph
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    N, T, mbs = 8192, 32, 1024
    rows = N * T
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.rand((rows, 480), device=dev, generator=g) < 0.1
    mask = torch.rand((rows, 38), device=dev, generator=g) < 0.5
    mask[:, 0] = True
    action = torch.zeros(rows, dtype=torch.int32, device=dev)
    flat = Transition(torch.zeros(rows, dtype=torch.bool, device=dev), action, torch.randn(rows, device=dev, generator=g) * 0.1,
                      torch.randn(rows, device=dev, generator=g) * 0.1, -torch.rand(rows, device=dev, generator=g) - 0.5, obs, mask)
    adv, tgt = torch.randn(rows, device=dev, generator=g) * 0.1, torch.randn(rows, device=dev, generator=g) * 0.1
    fp = make_forward_pass("relu", "DeepMind")
    fms, nets = [], []
    for v in variants:
        cfg = dict(DEFAULTS, num_envs=N, num_steps=T, minibatch_size=mbs, update_epochs=1, lr=1e-5)
        cfg.update(parse(v))
        net = fp.init(0, device=dev)
        opt = make_optimizer(cfg, net)["opt"]
        fms.append(FusedMinibatch(cfg, net, opt, mbs, dev))
        nets.append(net)
    times = [[] for _ in variants]
    perm = torch.randperm(rows, device=dev, generator=g)
    for r in range(opts["rounds"] + 1):
        for i, fm in enumerate(fms):
            fm.begin_update(flat, adv, tgt, [perm])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fm.run_steps(opts["steps"])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            fm.end_update()
            if r:
                times[i].append(dt / opts["steps"] * 1e3)
    for v, ts in zip(variants, times):
        print(f"{v:40s} {np.median(ts):.4f} ms per minibatch (min {min(ts):.4f}, max {max(ts):.4f}, {len(ts)} rounds of {opts['steps']} steps)")
    if opts["check"] and len(nets) > 1:
        p0 = torch.cat([p.detach().reshape(-1) for p in nets[0].parameters()])
        for v, n in zip(variants[1:], nets[1:]):
            p = torch.cat([q.detach().reshape(-1) for q in n.parameters()])
            print(f"max |param({variants[0]}) - param({v})| = {float((p0 - p).abs().max()):.3e}  (mean {float((p0 - p).abs().mean()):.3e}; "
                  f"lr = 1e-5, {opts['steps'] * (opts['rounds'] + 1)} steps)")


if __name__ == "__main__":
    main()
