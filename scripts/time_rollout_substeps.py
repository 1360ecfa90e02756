"""Random-policy rollout at 8192 x 32 with 1 / 4 env.step calls per macro-step (normal_step / the competitive macro-step with every
seat random, src/utils.py:69-128): the library default (k_rollout_fs at substeps 1, k_rollout_ws at 4) and k_rollout_ws forced
(BRL_ROLLOUT_FS=0); 100 launches between one HIP-event pair each (3 rotating Transition buffers)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import brl_amd  # noqa: E402
from brl_amd import _capi  # noqa: E402
from brl_amd.roll_out import alloc_transition  # noqa: E402

keys, values = bench.synthetic_lut(100_000, 0)
n, T = 8192, 32
dev = torch.device("cuda", 0)
trajs = [alloc_transition(T, n, dev) for _ in range(3)]
ptrs = []
for tr in trajs:
    p = _capi.TransitionPtrs()
    for name in _capi.TransitionPtrs._names:
        setattr(p, name, getattr(tr, name).data_ptr())
    ptrs.append(p)
lo, lm = torch.empty((n, 480), dtype=torch.bool, device=dev), torch.empty((n, 38), dtype=torch.bool, device=dev)
tc = torch.zeros(1, dtype=torch.int64, device=dev)
for fs in ("1", "0"):
    os.environ["BRL_ROLLOUT_FS"] = fs
    env = brl_amd.BridgeBidding(lut=(keys, values), device=dev)
    st = env.init(0, num_envs=n)
    L, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    for sub in (1, 4):
        def launch(i):
            _capi.check(L.brl_rollout_random(env._h, st.packed.data_ptr(), n, T, sub, (i * T * sub) & 0xFFFFFFFF, 7600.0, C.byref(ptrs[i % 3]),
                                             lo.data_ptr(), lm.data_ptr(), tc.data_ptr(), s))
        for i in range(20):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(100):
            launch(20 + i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        print(f"{'default      ' if fs == '1' else 'BRL_ROLLOUT_FS=0'} substeps {sub}: {us:7.2f} us per 8192 x 32 rollout = "
              f"{n * T / us / 1e3:.2f} G macro-steps/s = {n * T * sub / us / 1e3:.2f} G raw env-steps/s")
