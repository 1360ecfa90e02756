"""Launches the fused rollout a few times (target of rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from brl_amd import _capi as _c0
if os.environ.get('LIB'):
    _c0.LIB_PATH = os.environ['LIB']
import brl_amd
from brl_amd import _capi
from brl_amd.roll_out import alloc_transition
from brl_amd.bridge_bidding import _stream
from bench import synthetic_lut
N, T = 8192, 32
keys, values = synthetic_lut(100000, 0)
env = brl_amd.BridgeBidding(lut=(keys, values))
traj = alloc_transition(T, N, env.device)
st = env.init(0, num_envs=N)
p = _capi.TransitionPtrs()
for f in _capi.TransitionPtrs._names:
    setattr(p, f, _capi.ptr(getattr(traj, f)))
for i in range(6):
    _capi.check(_capi.lib().brl_rollout_random(env._h, _capi.ptr(st.packed), N, T, 1, i * T, 7600.0, C.byref(p), None, None, None, _stream()))
torch.cuda.synchronize()
