import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brl_amd import _capi as _c0
if os.environ.get('LIB'):
    _c0.LIB_PATH = os.environ['LIB']
import brl_amd
from oracle import Oracle
d = np.load("tests/golden/wb5_dds_1000.npz")
n, T = int(os.environ.get("N", 2048)), int(os.environ.get("T", 32))
SUB = int(os.environ.get("SUB", 1))
env = brl_amd.BridgeBidding(lut=(d["keys"], d["values"]))
orc = Oracle(d["keys"], d["values"])
roll = brl_amd.make_random_roll_out({"num_steps": T, "game_mode": "competitive" if SUB == 4 else "normal"}, env)
import brl_amd.roll_out as R
st = env.init(2024, num_envs=n)
ref = orc.init_random(n, seed=2024)
from brl_amd import _capi
import ctypes as C
from brl_amd.bridge_bidding import _stream
traj = R.alloc_transition(T, n, env.device)
p = _capi.TransitionPtrs()
for f in _capi.TransitionPtrs._names:
    setattr(p, f, _capi.ptr(getattr(traj, f)))
_capi.check(_capi.lib().brl_rollout_random(env._h, _capi.ptr(st.packed), n, T, SUB, 0, 7600.0, C.byref(p), None, None, None, _stream()))
want = orc.rollout_random(ref, T, seed=2024, substeps=SUB)
torch.cuda.synchronize()
for name in ("obs", "legal_action_mask", "action", "done", "reward", "log_prob"):
    g = getattr(traj, name).cpu().numpy(); g = g.astype(np.uint8) if g.dtype == np.bool_ else g
    o = want[name]
    bad = (g != o)
    print(name, "mismatch elems", int(bad.sum()))
    if bad.any():
        idx = np.argwhere(bad)
        print("  steps:", np.unique(idx[:, 0])[:40])
        print("  tables (first 40):", np.unique(idx[:, 1])[:40], "count", len(np.unique(idx[:, 1])))
        if bad.ndim == 3:
            print("  cols:", np.unique(idx[:, 2])[:60])
