import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brl_amd
from oracle import Oracle
d = np.load("tests/golden/wb5_dds_1000.npz")
n, T = int(os.environ.get("N", 2048)), 32
env = brl_amd.BridgeBidding(lut=(d["keys"], d["values"]))
orc = Oracle(d["keys"], d["values"])
roll = brl_amd.make_random_roll_out({"num_steps": T}, env)
st = env.init(2024, num_envs=n)
ref = orc.init_random(n, seed=2024)
rs, traj = roll((None, None, st, None, 0, 0))
want = orc.rollout_random(ref, T, seed=2024)
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
