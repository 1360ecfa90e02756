"""Probe: ONE bf16 graph rollout of 8192 tables against TWO independent rollouts of 4096 tables replayed on two streams at the same
time (the ~2 us between consecutive graph nodes and each launch's prologue / epilogue of one chain could hide behind the other
chain's kernels).  Prints ms per 8192 tables x 32 macro-steps for both forms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brl_amd
from brl_amd.models import make_forward_pass
from brl_amd.roll_out import _PolicyRollout
from brl_amd.train import DEFAULTS
from bench import synthetic_lut

T = 32
dt = torch.bfloat16
fp = make_forward_pass("relu", "DeepMind")
params, opp = fp.init(0, device="cuda"), fp.init(1, device="cuda")
lut = synthetic_lut(100000, 0)


def engine(n, env_offset):
    env = brl_amd.BridgeBidding(lut=lut, env_offset=env_offset)
    eng = _PolicyRollout(env, n, T, 7600.0, "competitive", True, dt, fp, fp, True)
    st = env.init(0, num_envs=n)
    rs = (params, None, st, st.observation, 0, 0)
    eng.run(rs, opp)          # capture
    eng.run(rs, opp)
    torch.cuda.synchronize()
    return eng, rs


def timed(f, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


one, rs1 = engine(8192, 0)
print("one chain of 8192 tables: %.2f ms" % timed(lambda: [g.replay() for g in one.graphs]))
a, rsa = engine(4096, 0)
b, rsb = engine(4096, 4096)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    for ga, gb in zip(a.graphs, b.graphs):
        with torch.cuda.stream(sa):
            ga.replay()
        with torch.cuda.stream(sb):
            gb.replay()
    cur.wait_stream(sa); cur.wait_stream(sb)


print("two chains of 4096 tables, two streams: %.2f ms" % timed(both))
print("one chain of 4096 tables alone: %.2f ms" % timed(lambda: [g.replay() for g in a.graphs]))
