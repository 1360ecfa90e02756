import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brl_amd import _capi
if os.environ.get("LIB"):
    _capi.LIB_PATH = os.environ["LIB"]
import numpy as np, torch, ctypes as C
import brl_amd
from brl_amd.roll_out import alloc_transition
from brl_amd.bridge_bidding import _stream
from bench import synthetic_lut
N, T = int(os.environ.get('N', 8192)), 32
keys, values = synthetic_lut(100000, 0)
res = {}
for cfg in os.environ.get("CFGS", "32x11").split(","):
    os.environ["BRL_ROLLOUT_WS"] = cfg
    env = brl_amd.BridgeBidding(lut=(keys, values))
    traj = alloc_transition(T, N, env.device)
    st = env.init(0, num_envs=N)
    p = _capi.TransitionPtrs()
    for f in _capi.TransitionPtrs._names:
        setattr(p, f, _capi.ptr(getattr(traj, f)))
    def launch(d):
        _capi.check(_capi.lib().brl_rollout_random(env._h, _capi.ptr(st.packed), N, T, 1, d, 7600.0, C.byref(p), None, None, None, _stream()))
    for i in range(10): launch(i*T)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    torch.cuda.synchronize()
    for i,(a,b) in enumerate(evs):
        a.record(); launch((10+i)*T); b.record()
    torch.cuda.synchronize()
    res[cfg] = round(float(np.median([a.elapsed_time(b) for a,b in evs]))*1e3,1)
print(os.environ.get("LIB","default"), json.dumps(res))
