"""A few ppo.py iterations at configs[3]'s size through brl_amd.train.train (evaluators on, as in the reference's loop) and the
seconds each phase took: eval_s (ppo.py:366-381,461 — every evaluation in front of the rollout), rollout_s, gae_s, update_s."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brl_amd.train import train, DEFAULTS

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(num_envs=8192, num_steps=32, minibatch_size=1024, update_epochs=10, total_timesteps=8192 * 32 * iters,
           graph_rollout=True, evaluate=True, save_model=False, log_path=tempfile.mkdtemp(), exp_name="phases")
cfg.update(dict(a.split("=", 1) for a in sys.argv[2:]))
for k, v in list(cfg.items()):
    if isinstance(v, str) and k in DEFAULTS and isinstance(DEFAULTS[k], (int, float, bool)) and not isinstance(DEFAULTS[k], str):
        cfg[k] = type(DEFAULTS[k])(int(v)) if isinstance(DEFAULTS[k], bool) else type(DEFAULTS[k])(v)
t0 = time.perf_counter()
_, hist = train(cfg, log=lambda s: None)
for r in hist:
    print({k: round(r[k], 4) for k in ("eval_s", "rollout_s", "gae_s", "update_s")}, "macro-steps/s %.0f" % r["macro_steps_per_s"])
print("wall %.1f s for %d iterations" % (time.perf_counter() - t0, len(hist)))
