"""Regenerates brl_amd/tuned/tunableop_gfx950.csv: runs the policy path once (configs[3] phases, fp32 and bf16 rollout forwards,
duplicate evaluation) with torch TunableOp tuning every GEMM shape it meets.  GPU box only; ~1-2 minutes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brl_amd import tuned  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else tuned.PATH
if os.path.exists(out):
    os.remove(out)
torch.cuda.tunable.set_max_tuning_duration(50)
torch.cuda.tunable.set_max_tuning_iterations(200)
assert tuned.enable(tuning=True, path=out)
import bench  # noqa: E402

res = bench.bench_secondary(torch, torch.device("cuda", 0))
print({k: v for k, v in res["config3"]["update"].items() if k in ("ms", "ms_per_minibatch", "gemm_tflops")})
if hasattr(torch.cuda.tunable, "write_file"):
    torch.cuda.tunable.write_file(out)
else:   # this torch writes the file when the process ends: print what it holds
    for r in torch.cuda.tunable.get_results():
        print(r)
