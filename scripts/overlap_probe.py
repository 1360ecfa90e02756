"""What does a collective's kernel cost the GEMMs it overlaps?  (the first question for a node run.)

The multi-rank PPO step (brl_amd.fused_update.FusedMinibatch, world-8 geometry, collectives inside the eight-step hipGraph) on ONE GPU
with every collective replaced by a SPIN kernel on the collective stream: `channels` workgroups that hold their CUs for the time the
collective would take at world 8 (ring per link ~ 153 GB/s + ~10 us: 4.2 MB bucket ~ 34 us, 2.1 MB ~ 22 us, 4 KB ~ 12 us, the flat
14.7 MB all-reduce ~ 180 us).  Models CU occupancy and duration, not the HBM / xGMI traffic.  Prints ms per minibatch step for
sharded / flat x channels in {0 (stream edges only), 8, 16, 32, 64}.

    hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o scripts/micro/libspin.so scripts/micro/spin_kernel.hip
    python scripts/overlap_probe.py [out_file]
"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brl_amd.models import make_forward_pass   # noqa: E402
from brl_amd.roll_out import Transition        # noqa: E402
from brl_amd.train import DEFAULTS             # noqa: E402
from brl_amd.update import FusedMinibatch, make_optimizer   # noqa: E402

WORLD, RANK = 8, 3


def usec_of(nbytes, ring_factor):
    """ring over one xGMI link at ~153 GB/s: (W - 1) / W of the bytes per pass (all-reduce: two passes) + ~10 us"""
    return 10.0 + ring_factor * (WORLD - 1) / WORLD * nbytes / 153e9 * 1e6


def main(out_path):
    so = os.path.join(ROOT, "scripts", "micro", "libspin.so")
    if not os.path.exists(so):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", so,
                               os.path.join(ROOT, "scripts", "micro", "spin_kernel.hip")])
    spin = ctypes.CDLL(so)
    spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    N, T, mbs = 8192, 32, 1024
    rows = N * T
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.rand((rows, 480), device=dev, generator=g) < 0.1
    mask = torch.rand((rows, 38), device=dev, generator=g) < 0.5
    mask[:, 0] = True
    flat = Transition(torch.zeros(rows, dtype=torch.bool, device=dev), torch.zeros(rows, dtype=torch.int32, device=dev),
                      torch.randn(rows, device=dev, generator=g) * 0.1, torch.randn(rows, device=dev, generator=g) * 0.1,
                      -torch.rand(rows, device=dev, generator=g) - 0.5, obs, mask)
    adv, tgt = torch.randn(rows, device=dev, generator=g) * 0.1, torch.randn(rows, device=dev, generator=g) * 0.1
    fp = make_forward_pass("relu", "DeepMind")
    side = torch.cuda.Stream()
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    class _Work:                       # like a collective's Work: waits for ITS collective's end, not for the whole stream
        def __init__(self):
            self.done = torch.cuda.Event()
            self.done.record(side)

        def wait(self):
            torch.cuda.current_stream().wait_event(self.done)

    class SpinCollectives:
        capturable = True

        def __init__(self, channels):
            self.rank, self.world, self.channels, self.total_us = RANK, WORLD, channels, 0.0

        def _run(self, usec, async_op):
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            if self.channels:
                assert spin.spin_launch(self.channels, usec, ctypes.c_void_p(side.cuda_stream)) == 0
            self.total_us += usec
            if async_op:
                return _Work()
            cur.wait_stream(side)
            return None

        def all_reduce(self, t, async_op):
            return self._run(usec_of(t.numel() * 4, 2.0), async_op)

        def reduce_scatter(self, out, inp, async_op):
            return self._run(usec_of(inp.numel() * 4, 1.0), async_op)

        def all_gather(self, out, inp, async_op):
            return self._run(usec_of(out.numel() * 4, 1.0), async_op)

    cfg0 = dict(DEFAULTS, num_envs=N, num_steps=T, minibatch_size=mbs, update_epochs=1, lr=1e-5)
    say(f"one rank's PPO minibatch step, world-{WORLD} geometry, collectives = spin kernels of `channels` workgroups on the collective "
        f"stream, inside the eight-step hipGraph; 256 steps, median of 3")
    for mode in ("sharded", "flat"):
        for channels in (0, 8, 16, 32, 64):
            net = fp.init(0, device=dev)
            opt = make_optimizer(cfg0, net)["opt"]
            co = SpinCollectives(channels)
            fm = FusedMinibatch(dict(cfg0, grad_allreduce=mode), net, opt, mbs, dev, world=WORLD, collective=co)
            co.total_us = 0.0
            fm._run_program  # noqa: B018
            ts = []
            perm = torch.randperm(rows, device=dev, generator=g)
            for rep in range(4):
                fm.begin_update(flat, adv, tgt, [perm])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fm.run_steps(256)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 256 * 1e3)
                fm.end_update()
            per_step = sum(usec_of(x, f) for x, f in _collective_bytes(fm, mode))
            say(f"{mode:8s} channels {channels:3d}: {np.median(ts[1:]):.4f} ms per step  (collective time per step if fully exposed: {per_step:.0f} us)")
            del fm, net, opt
    if out_path:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        open(out_path, "w").write("\n".join(lines) + "\n")


def _collective_bytes(fm, mode):
    if mode == "flat":
        return [(fm.n * 4, 2.0)]
    out = []
    for ln in fm.bucket_len:
        out += [(ln * WORLD * 4, 1.0), (ln * WORLD * 4, 1.0)]      # reduce-scatter + all-gather of the bucket
    return out + [(fm.norm_partials.numel() * 4, 1.0)]


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/overlap_probe.txt")
