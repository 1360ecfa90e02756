"""brl_amd.train at configs[3]'s size for N iterations (evaluators on): device memory in use / reserved and the iteration time as
the loop goes on — a leak (graph pools, cached watches, snapshots) or a slow-down would show here.  usage: python scripts/soak_train.py [iterations] [key=value ...]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from brl_amd.train import train

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = dict(num_envs=8192, num_steps=32, minibatch_size=1024, update_epochs=10, total_timesteps=8192 * 32 * iters, graph_rollout=True,
           evaluate=True, save_model=False, log_path=tempfile.mkdtemp(), exp_name="soak")
cfg.update(dict(a.split("=", 1) for a in sys.argv[2:]))     # e.g. actor_model_type=FAIR
t0 = time.perf_counter()
mem = []


def log(line):   # (called once per iteration)
    mem.append((torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20))


_, hist = train(cfg, log=log)
torch.cuda.synchronize()
for i, r in enumerate(hist):
    if i % max(1, iters // 8) == 0 or i == len(hist) - 1:
        print(f"iteration {i:4d}: {1e3 * (r['eval_s'] + r['rollout_s'] + r['gae_s'] + r['update_s']):7.1f} ms  "
              f"(eval {1e3 * r['eval_s']:.1f}, rollout {1e3 * r['rollout_s']:.1f}, update {1e3 * r['update_s']:.1f})  "
              f"allocated {mem[i][0]:.0f} MiB, reserved {mem[i][1]:.0f} MiB; loss {r['train/total_loss']:.4f} value {r['train/value_loss']:.4f} "
              f"entropy {r['train/policy_entropy']:.3f} kl {r['train/approx_kl']:.5f}")
print(f"wall {time.perf_counter() - t0:.1f} s for {len(hist)} iterations; device memory allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB, "
      f"reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB, peak allocated {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB")
