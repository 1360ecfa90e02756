"""Target of rocprofv3 --kernel-trace --stats: a few bf16 policy-in-the-loop rollouts (competitive mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brl_amd
from brl_amd.models import make_forward_pass
from brl_amd.train import DEFAULTS
from bench import synthetic_lut
N, T = 8192, 32
dt = os.environ.get("DT", "bf16")
env = brl_amd.BridgeBidding(lut=synthetic_lut(100000, 0))
fp = make_forward_pass("relu", "DeepMind")
params, opp = fp.init(0, device="cuda"), fp.init(1, device="cuda")
cfg = dict(DEFAULTS, num_envs=N, num_steps=T, inference_dtype=None if dt == "fp32" else dt,
           graph_rollout=bool(int(os.environ.get("GRAPH", "0"))))
roll = brl_amd.make_roll_out(cfg, env, fp, fp)
st = env.init(0, num_envs=N)
rs = (params, None, st, st.observation, 0, 0)
roll(rs, opp); roll(rs, opp); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): roll(rs, opp)
torch.cuda.synchronize()
print("rollout %s graph=%s: %.2f ms" % (dt, os.environ.get("GRAPH", "0"), (time.perf_counter() - t0) / 5 * 1e3))
