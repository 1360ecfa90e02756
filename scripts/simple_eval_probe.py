"""make_simple_evaluate at num_eval_envs = 10000 (ppo.py:366): the by-turn loop against the macro-step loop (BRL_SIMPLE_EVAL_BY_TURN=0)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import brl_amd
from bench import LUT_LEN, synthetic_lut
from brl_amd.evaluation import make_simple_evaluate
from brl_amd.models import make_forward_pass

dev = torch.device("cuda:0")
env = brl_amd.BridgeBidding(lut=synthetic_lut(LUT_LEN, 0), device=dev)
fp = make_forward_pass("relu", "DeepMind")
a, o = fp.init(0, device=dev), fp.init(1, device=dev)
ev = make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", o, 10000)
for mode in ("1", "0"):
    os.environ["BRL_SIMPLE_EVAL_BY_TURN"] = mode
    for _ in range(3):
        r = ev(a, 5)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        r = ev(a, 5)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"BRL_SIMPLE_EVAL_BY_TURN={mode}: {np.median(ts) * 1e3:.2f} ms (min {min(ts) * 1e3:.2f}), mean return {float(r):.4f}")
