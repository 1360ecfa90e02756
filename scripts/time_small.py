import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brl_amd
from brl_amd.gae import gae_scan
from bench import synthetic_lut
keys, values = synthetic_lut(100000, 0)
env = brl_amd.BridgeBidding(lut=(keys, values))
T, N = 32, 8192
done = torch.rand(T, N, device="cuda") < 0.1
value = torch.randn(T, N, device="cuda"); reward = torch.randn(T, N, device="cuda"); last = torch.randn(N, device="cuda")
st = env.init(0, num_envs=N)
def timeit(fn, n=100):
    for _ in range(10): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return round(float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3, 1)
print(json.dumps({"gae_us": timeit(lambda: gae_scan(env, done, value, reward, last, 1.0, 0.95)),
                  "observe_us": timeit(lambda: env.observe(st))}))
