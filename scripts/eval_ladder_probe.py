"""configs[2] (8192-board duplicate evaluation, two DeepMind MLPs, fp32) iteration by iteration: rows forwarded (the batch-size
ladder of brl_amd/evaluation.py::_ActiveRows), GPU time between the iterations' ends (events) and host time per iteration.
usage (GPU box): python scripts/eval_ladder_probe.py [out.txt]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import brl_amd
from bench import LUT_LEN, NUM_ENVS, synthetic_lut
from brl_amd import evaluation as ev
from brl_amd.models import make_forward_pass


def main():
    dev = torch.device("cuda:0")
    keys, values = synthetic_lut(LUT_LEN, 0)
    env = brl_amd.BridgeBidding(lut=(keys, values), device=dev)
    fp = make_forward_pass("relu", "DeepMind")
    t1, t2 = fp.init(0, device=dev), fp.init(1, device=dev)
    dup = ev.make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", NUM_ENVS)
    for _ in range(2):
        dup(t1, t2, 123)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dup(t1, t2, 123)
    torch.cuda.synchronize()
    plain = time.perf_counter() - t0
    log = []
    orig = ev._ActiveRows.forward

    def probe(self, fwd, obs, env_, x=None):
        r = orig(self, fwd, obs, env_, x)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        log.append((self.m if self.idx is not None else self.n, e, time.perf_counter()))
        return r

    ev._ActiveRows.forward = probe
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    h0 = time.perf_counter()
    dup(t1, t2, 123)
    torch.cuda.synchronize()
    total = time.perf_counter() - h0
    ev._ActiveRows.forward = orig
    out = [f"duplicate evaluation, {NUM_ENVS} boards: {plain * 1e3:.2f} ms unprobed, {total * 1e3:.2f} ms probed, {len(log)} iterations",
           "iter  rows  gpu_us(since previous forward's end)  host_us"]
    prev_e, prev_h = start, h0
    by_m = {}
    for i, (m, e, h) in enumerate(log):
        g = prev_e.elapsed_time(e) * 1e3
        out.append(f"{i:4d} {m:5d} {g:9.1f} {(h - prev_h) * 1e6:9.1f}")
        by_m.setdefault(m, []).append(g)
        prev_e, prev_h = e, h
    out.append("rows: iterations, total gpu ms, mean us")
    for m in sorted(by_m, reverse=True):
        v = by_m[m]
        out.append(f"{m:5d}: {len(v):3d} {sum(v) / 1e3:7.2f} {np.mean(v):8.1f}")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
