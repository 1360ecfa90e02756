"""Target of rocprofv3 --kernel-trace --stats: the simple duplicate evaluator at num_eval_envs = 8192 (BASELINE configs[2])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brl_amd
from brl_amd.models import make_forward_pass
from brl_amd.evaluation import make_simple_duplicate_evaluate
from bench import synthetic_lut
N = 8192
env = brl_amd.BridgeBidding(lut=synthetic_lut(100000, 0))
fp = make_forward_pass("relu", "DeepMind")
p1, p2 = fp.init(0, device="cuda"), fp.init(1, device="cuda")
ev = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", N)
if os.environ.get("BRL_TUNED_GEMM", "0") == "1":   # as inside a training process, where the rollout / update have enabled the lookups
    from brl_amd import tuned
    tuned.enable()
for i in range(3): ev(p1, p2, 100 + i)   # (new batch sizes cost host time once: the library's per-shape heuristics)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5): ev(p1, p2, i + 1)
torch.cuda.synchronize()
print("duplicate evaluation of %d boards: %.2f ms" % (N, (time.perf_counter() - t0) / 5 * 1e3))
