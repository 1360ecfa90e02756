"""The step's clip + Adam launches (k_adam_norm_fin + k_adam_apply: 103 MB of traffic over a 59 MB working set) back to back in a
graph, nothing else between them: does the sweep run faster when its buffers are the only thing the memory-side cache sees?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from brl_amd.models import make_forward_pass
from brl_amd.train import DEFAULTS
from brl_amd.update import FusedMinibatch, make_optimizer
from brl_amd._capture import quiet_gc

dev = torch.device("cuda:0")
fp = make_forward_pass("relu", "DeepMind")
net = fp.init(0, device=dev)
cfg = dict(DEFAULTS, num_envs=8192, num_steps=32, minibatch_size=1024, update_epochs=1, lr=1e-6)
fm = FusedMinibatch(cfg, net, make_optimizer(cfg, net)["opt"], 1024, dev)
fm.G.normal_(0, 1e-3)
K = 64
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    fm._opt()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with quiet_gc(), torch.cuda.graph(g), torch.no_grad():
    for _ in range(K):
        fm._opt()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (10 * K)
print(f"k_adam_norm_fin + k_adam_apply alone, back to back: {dt * 1e6:.2f} us per pair (in the step: 4.8 + 19.1 = 23.9 us under rocprofv3)")
