"""How much of the bench step (rollout + GAE, two launches) is launch gap?  Times eager launches vs one hipGraph
replay of the same two kernels (fixed draw_base in the captured graph: a probe, not the bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brl_amd
from brl_amd.gae import gae_scan
from brl_amd.roll_out import alloc_transition
from bench import synthetic_lut, NUM_ENVS, NUM_STEPS, LUT_LEN
dev = torch.device("cuda", 0)
keys, values = synthetic_lut(LUT_LEN, 0)
env = brl_amd.BridgeBidding(lut=(keys, values), device=dev)
roll = brl_amd.make_random_roll_out({"num_steps": NUM_STEPS, "reward_scale": 7600, "return_last_obs": True}, env)
state = env.init(0, num_envs=NUM_ENVS)
traj = alloc_transition(NUM_STEPS, NUM_ENVS, dev)
last_val = torch.zeros(NUM_ENVS, dtype=torch.float32, device=dev)
rs = (None, None, state, None, 0, 0)
def step(rs):
    rs, tb = roll(rs, out=traj)
    gae_scan(env, tb.done, tb.value, tb.reward, last_val, 1.0, 0.95)
    return rs
for _ in range(20): rs = step(rs)
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K): rs = step(rs)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / K * 1e6
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): rs = step(rs)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    rs = step(rs)
for _ in range(20): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / K * 1e6
print(f"eager {eager:.1f} us/step   graph replay {graph:.1f} us/step")
