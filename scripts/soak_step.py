"""Randomised differential soak of the per-step API against the CPU oracle: env.step with and without auto-reset,
legal and (rarely) illegal actions, duplicate_step pairs, observe for every seat — changing sizes, seeds and
tables-per-wave.  Run under `timeout`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import brl_amd
from oracle import Oracle
from bench import synthetic_lut
from gpu_util import assert_state_equal, random_legal_actions, to_np
budget = float(os.environ.get("SOAK_SECONDS", "60"))
keys, values = synthetic_lut(5000, 11)
orc = Oracle(keys, values)
rng = np.random.default_rng(5)
t_end = time.time() + budget
steps = 0
while time.time() < t_end:
    k = int(rng.choice([1, 2, 4, 8]))
    os.environ["BRL_TABLES_PER_WAVE"] = str(k)
    env = brl_amd.BridgeBidding(lut=(keys, values))
    n = int(rng.choice([1, 3, 64, 257, 1000]))
    seed = int(rng.integers(1 << 30))
    autoreset = bool(rng.integers(2))
    p_illegal = float(rng.choice([0.0, 0.0, 0.002]))
    st = env.init(seed, num_envs=n)
    ref = orc.init_random(n, seed=seed)
    assert_state_equal(st, ref, where="init")
    for t in range(int(rng.integers(20, 400))):
        act = random_legal_actions(rng, ref["legal_action_mask"])
        bad = rng.random(n) < p_illegal
        act = np.where(bad, rng.integers(0, 38, n), act).astype(np.int32)
        st = env.step(st, torch.from_numpy(act).to(env.device), autoreset=autoreset)
        orc.step(ref, act, autoreset=autoreset, seed=seed)
        steps += 1
        if t % 7 == 0:
            assert_state_equal(st, ref, where=f"K={k} n={n} seed={seed} autoreset={autoreset} step {t}")
        if t % 23 == 0:
            pid = rng.integers(0, 4, n).astype(np.int32)
            got = to_np(brl_amd._observe(st, torch.from_numpy(pid).to(env.device)))
            assert np.array_equal(got, orc.observe(ref, pid)), "observe"
        if not autoreset and ref["terminated"].all():
            break
    assert_state_equal(st, ref, where="final")
print(f"soak_step ok: {steps} batched steps")
