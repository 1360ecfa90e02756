"""ms per PPO minibatch step of the FAIR network (src/models.py:34-69) at configs[3]'s sizes (8192 x 32 synthetic trajectory, minibatch
1024, one epoch = 256 steps), through make_update_step: whichever path update_step picks (the record names it)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brl_amd.models import make_forward_pass   # noqa: E402
from brl_amd.roll_out import Transition        # noqa: E402
from brl_amd.train import DEFAULTS             # noqa: E402
from brl_amd.update import make_optimizer, make_update_step   # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    N, T, mbs = 8192, 32, 1024
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.rand((T, N, 480), device=dev, generator=g) < 0.1
    mask = torch.rand((T, N, 38), device=dev, generator=g) < 0.5
    mask[..., 0] = True
    traj = Transition(torch.zeros((T, N), dtype=torch.bool, device=dev), torch.zeros((T, N), dtype=torch.int32, device=dev),
                      torch.randn((T, N), device=dev, generator=g) * 0.1, torch.randn((T, N), device=dev, generator=g) * 0.1,
                      -torch.rand((T, N), device=dev, generator=g) - 0.5, obs, mask)
    adv, tgt = torch.randn((T, N), device=dev, generator=g) * 0.1, torch.randn((T, N), device=dev, generator=g) * 0.1
    for variant in sys.argv[1:] or ["fused_update=1", "fused_update=0"]:
        cfg = dict(DEFAULTS, num_envs=N, num_steps=T, minibatch_size=mbs, update_epochs=1, lr=1e-5,
                   **{kv.split("=")[0]: bool(int(kv.split("=")[1])) for kv in variant.split(",")})
        for act in ("relu",):
            fp = make_forward_pass(act, "FAIR")
            net = fp.init(0, device=dev)
            upd = make_update_step(cfg, fp)
            rs = (net, make_optimizer(cfg, net), None, None, 0, 5)
            ts = []
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for rep in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    rs, info = upd(rs, traj, adv, tgt)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
            path = type(rs[1].get("graphed")).__name__ if rs[1].get("graphed") else "eager"
            print(f"FAIR {act} {variant:18s} path {path:18s} {np.median(ts[1:]) / 256 * 1e3:.4f} ms per minibatch step "
                  f"(update {np.median(ts[1:]) * 1e3:.1f} ms; loss {float(info[0][-1][-1]):.5f})", flush=True)


if __name__ == "__main__":
    main()
