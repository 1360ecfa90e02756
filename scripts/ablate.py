"""Times k_rollout_random variants (which outputs are requested) to see where the time goes."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import brl_amd
from brl_amd import _capi
from brl_amd.roll_out import alloc_transition, Transition
from brl_amd.bridge_bidding import _stream
from bench import synthetic_lut

N, T = 8192, 32
keys, values = synthetic_lut(100000, 0)
res = {}
for K in (os.environ.get("KS", "k4,16x4,16x6,16x10,32x6,32x8,32x10,32x16,64x10,64x16").split(",")):
    if K.startswith("k"):
        os.environ["BRL_TABLES_PER_WAVE"] = K[1:]
        os.environ["BRL_ROLLOUT_WS"] = "0"
    else:
        os.environ["BRL_ROLLOUT_WS"] = K
    env = brl_amd.BridgeBidding(lut=(keys, values))
    traj = alloc_transition(T, N, env.device)
    variants = {
        "full": traj,
        "no_obs": traj._replace(obs=None),
        "no_obs_mask": traj._replace(obs=None, legal_action_mask=None),
        "obs_only": Transition(None, None, None, None, None, traj.obs, None),
        "nothing": Transition(None, None, None, None, None, None, None),
    }
    for name, tr in variants.items():
        st = env.init(0, num_envs=N)
        p = _capi.TransitionPtrs()
        for f in _capi.TransitionPtrs._names:
            setattr(p, f, _capi.ptr(getattr(tr, f)))
        def launch(d):
            _capi.check(_capi.lib().brl_rollout_random(env._h, _capi.ptr(st.packed), N, T, 1, d, 7600.0, C.byref(p), None, None, None, _stream()))
        for i in range(10):
            launch(i * T)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            a.record(); launch((10 + i) * T); b.record()
        torch.cuda.synchronize()
        res[f"K{K}_{name}"] = round(float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3, 1)
print(json.dumps(res, indent=1))
