"""cProfile of the 8192-board duplicate evaluation's HOST side (where does the Python time go).  usage: python scripts/eval_host_profile.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import brl_amd
from bench import LUT_LEN, NUM_ENVS, synthetic_lut
from brl_amd import evaluation as ev
from brl_amd.models import make_forward_pass

dev = torch.device("cuda:0")
env = brl_amd.BridgeBidding(lut=synthetic_lut(LUT_LEN, 0), device=dev)
fp = make_forward_pass("relu", "DeepMind")
t1, t2 = fp.init(0, device=dev), fp.init(1, device=dev)
dup = ev.make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", NUM_ENVS)
for _ in range(3):
    dup(t1, t2, 123)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    dup(t1, t2, 123)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
