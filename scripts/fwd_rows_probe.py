"""The evaluators' small-batch forward, per call: brl_mlp_forward_rows (one host call, own kernels) against the library path
(brl_obs_cast_rows + InferenceSnapshot.heads through torch / hipBLASLt + index_copy_), back to back (= max(host, GPU) per call).
usage (GPU box): python scripts/fwd_rows_probe.py [out.txt]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import brl_amd
from bench import synthetic_lut
from brl_amd import _capi
from brl_amd._capi import check, ptr
from brl_amd.bridge_bidding import _stream
from brl_amd.evaluation import _Forward
from brl_amd.models import make_forward_pass


def main():
    dev = torch.device("cuda:0")
    env = brl_amd.BridgeBidding(lut=synthetic_lut(1000, 0), device=dev)
    fp = make_forward_pass("relu", "DeepMind")
    net = fp.init(0, device=dev)
    fwd = _Forward(fp, net)
    n = 8192
    obs = torch.rand(n, 480, device=dev) < 0.1
    full = torch.zeros(n, 39, device=dev)
    out = []
    for m in (64, 256, 512, 1024, 2048, 4096):
        idx = torch.randperm(n, device=dev)[:m].contiguous()

        def own():
            fwd.rows(obs, idx, m, full, env)

        def lib():
            x = torch.empty((m, 480), dtype=torch.float32, device=dev)
            check(_capi.lib().brl_obs_cast_rows(env._h, ptr(obs), ptr(idx), m, ptr(x), 0, _stream()))
            full.index_copy_(0, idx, fwd(None, x))

        res = []
        for f in (own, lib):
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            reps = 300
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            host = time.perf_counter() - t0
            torch.cuda.synchronize()
            tot = time.perf_counter() - t0
            res.append((host / reps * 1e6, tot / reps * 1e6))
        out.append(f"m = {m:5d}: own {res[0][1]:7.1f} us per call (host {res[0][0]:6.1f})   library {res[1][1]:7.1f} us (host {res[1][0]:6.1f})")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
