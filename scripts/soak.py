"""Soak test of the fused rollout kernels: many back-to-back launches over changing seeds / sizes / step counts,
with periodic bit-exact checks against the CPU oracle.  Run under `timeout` (a hang is the failure mode looked for)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brl_amd
from oracle import Oracle
from bench import synthetic_lut
budget = float(os.environ.get("SOAK_SECONDS", "60"))
keys, values = synthetic_lut(20000, 3)
orc = Oracle(keys, values)
env = brl_amd.BridgeBidding(lut=(keys, values))
rng = np.random.default_rng(0)
t_end = time.time() + budget
launches = checks = 0
while time.time() < t_end:
    n = int(rng.choice([1, 31, 32, 33, 64, 96, 500, 2048, 4096, 8192]))   # multiples of 32 with substeps 1: k_rollout_fs (T > 40: in pieces)
    T = int(rng.choice([1, 7, 16, 32, 33, 40, 64]))
    sub = int(rng.choice([1, 1, 1, 4]))
    seed = int(rng.integers(1 << 30))
    env.set_rng(seed) if hasattr(env, "set_rng") else None
    roll = brl_amd.make_random_roll_out({"num_steps": T, "substeps": sub, "game_mode": "competitive" if sub == 4 else "normal"}, env)
    one_launch = sub == 1 and T <= 40 and n % 32 == 0 and rng.random() < 0.5   # rollout + calc_gae in one launch
    gamma, lam = float(rng.choice([1.0, 0.99, 0.9])), float(rng.choice([0.95, 1.0, 0.5]))
    if one_launch:
        roll_g = brl_amd.make_random_roll_out_with_gae({"num_steps": T, "gamma": gamma, "gae_lambda": lam}, env)
        lv = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(env.device)
    st = env.init(seed, num_envs=n)
    rs = (None, None, st, None, 0, 0)
    check = n <= 4096 and (checks < 400 or rng.random() < 0.03)  # keep checking (sparsely) for the whole run
    ref = orc.init_random(n, seed=seed) if check else None
    draw = 0
    for rep in range(int(rng.integers(1, 40))):
        if one_launch:
            rs, traj, adv, tgt = roll_g(rs, lv)
        else:
            rs, traj = roll(rs)
        launches += 1
        if check and rep < 3:
            want = orc.rollout_random(ref, T, seed=seed, substeps=sub, draw_base=draw)
            torch.cuda.synchronize()
            for name in ("obs", "legal_action_mask", "action", "done", "reward"):
                g = getattr(traj, name).cpu().numpy()
                assert np.array_equal(g.astype(want[name].dtype), want[name]), (name, n, T, sub, seed, rep)
            if one_launch:
                wa, wt = orc.gae(want["done"], want["value"], want["reward"], lv.cpu().numpy(), gamma, lam)
                assert np.array_equal(adv.cpu().numpy(), wa) and np.array_equal(tgt.cpu().numpy(), wt), ("gae", n, T, seed, rep)
            checks += 1
        draw += T * sub
    torch.cuda.synchronize()
print(f"soak ok: {launches} launches, {checks} oracle checks, BRL_ROLLOUT_FS={os.environ.get('BRL_ROLLOUT_FS', 'default')}")
