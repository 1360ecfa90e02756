"""Times the policy-in-the-loop phases (BASELINE configs 3/4) on one MI355X: rollout with the MLP in the
loop (fp32 / bf16 inference), GAE, PPO update, duplicate evaluation."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brl_amd
from brl_amd.models import make_forward_pass
from brl_amd.update import make_optimizer, make_update_step
from brl_amd.evaluation import make_simple_duplicate_evaluate
from brl_amd.train import DEFAULTS
from bench import synthetic_lut

N, T = 8192, 32
env = brl_amd.BridgeBidding(lut=synthetic_lut(100000, 0))
fp = make_forward_pass("relu", "DeepMind")
params, opp = fp.init(0, device="cuda"), fp.init(1, device="cuda")
res = {}
def sync_time(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out
for dt, graph in ((None, False), ("bf16", False), ("bf16", True)):
    cfg = dict(DEFAULTS, num_envs=N, num_steps=T, inference_dtype=dt, graph_rollout=graph)
    roll = brl_amd.make_roll_out(cfg, env, fp, fp)
    st = env.init(0, num_envs=N)
    rs = (params, None, st, st.observation, 0, 0)
    roll(rs, opp)  # (graph capture on the first call)
    t, (rs2, traj) = sync_time(lambda: roll(rs, opp), 5)
    tag = (dt or "fp32") + ("_graph" if graph else "")
    res[f"rollout_{tag}_ms"] = round(t * 1e3, 2)
    res[f"rollout_{tag}_macro_steps_per_s"] = round(N * T / t)
    res[f"rollout_{tag}_raw_env_steps_per_s"] = round(4 * N * T / t)
cfg = dict(DEFAULTS, num_envs=N, num_steps=T)
cfg["num_minibatches"] = N * T // cfg["minibatch_size"]; cfg["num_updates"] = 1
calc_gae = brl_amd.make_calc_gae(cfg, fp)
t, (adv, tgt) = sync_time(lambda: calc_gae(rs2, traj), 10)
res["calc_gae_ms"] = round(t * 1e3, 3)
upd = make_update_step(dict(cfg, update_epochs=1), fp)
opt_state = make_optimizer(cfg, params)
t, _ = sync_time(lambda: upd((params, opt_state, rs2[2], rs2[3], rs2[4], 0), traj, adv, tgt), 2)
res["update_one_epoch_ms"] = round(t * 1e3, 1)
res["update_per_minibatch_ms"] = round(t * 1e3 / cfg["num_minibatches"], 3)
eval_env = brl_amd.BridgeBidding(lut=synthetic_lut(100000, 1))
ev = make_simple_duplicate_evaluate(eval_env, "relu", "DeepMind", "relu", "DeepMind", N)
t, (info, _, _) = sync_time(lambda: ev(params, opp, 3), 2)
res["duplicate_eval_8192_boards_ms"] = round(t * 1e3, 1)
res["duplicate_eval_boards_per_s"] = round(N / t)
from brl_amd.evaluation import make_evaluate
ev2 = make_evaluate(eval_env, "relu", "DeepMind", "relu", "DeepMind", opp, N, duplicate=True)
t, _ = sync_time(lambda: ev2(params, 3), 2)
res["duplicate_eval_with_statistics_8192_boards_ms"] = round(t * 1e3, 1)
print(json.dumps(res))
