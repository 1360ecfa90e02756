"""Timing build (-DBRL_TIMING) of k_rollout_flow: per-wave s_memtime stamps (start, after the prologue barrier,
after every command slot, end) of a few workgroups, relative to the workgroup's first stamp."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libbrl_timing.so"
SRC = os.environ.get("SRC", os.path.join(ROOT, "brl_amd/csrc/brl_kernels.hip"))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                       "-DBRL_TIMING", "-I", os.path.join(ROOT, "include"), "-o", so, SRC], stderr=subprocess.DEVNULL)
os.environ["BRL_ROLLOUT_FLOW"] = "1"
from brl_amd import _capi
_capi.LIB_PATH = so
import numpy as np, torch, ctypes as C
import brl_amd
from brl_amd.roll_out import alloc_transition
from brl_amd.bridge_bidding import _stream
from bench import synthetic_lut
N, T = 8192, 32
keys, values = synthetic_lut(100000, 0)
tpb, nw = 32, 13
BRIEF = os.environ.get("BRIEF")
env = brl_amd.BridgeBidding(lut=(keys, values))
traj = alloc_transition(T, N, env.device)
st = env.init(0, num_envs=N)
p = _capi.TransitionPtrs()
for f in _capi.TransitionPtrs._names:
    setattr(p, f, _capi.ptr(getattr(traj, f)))
nblk = (N + tpb - 1) // tpb
dump = torch.zeros(nblk * nw * 48, dtype=torch.int64, device=env.device)
for i in range(5):
    dump.zero_()
    _capi.check(_capi.lib().brl_rollout_random(env._h, _capi.ptr(st.packed), N, T, 1, i * T, 7600.0, C.byref(p), None, None, _capi.ptr(dump), _stream()))
torch.cuda.synchronize()
d = dump.cpu().numpy().reshape(nblk, nw, 48)
t0 = d[:, :, 0].min()
print("ticks of s_memtime; kernel-wide span:", int(d[:, :, 47].max() - t0))
starts = d[:, :, 0].min(1) - t0
ends = d[:, :, 47].max(1) - t0
print("workgroup start spread: min %d max %d ; end: min %d max %d ; per-workgroup duration mean %d" % (
    starts.min(), starts.max(), ends.min(), ends.max(), (ends - starts).mean()))
names = ["logic", "prep", "apply", "scorer", "loader"] + ["emit%d" % i for i in range(nw - 5)]
for wg in ([] if BRIEF else [int(x) for x in os.environ.get("WGS", "5,130").split(",")]):
    b0 = d[wg, :, 0].min()
    print("workgroup", wg, "(start +%d)" % (b0 - t0))
    for w in range(nw):
        row = d[wg, w]
        vals = [int(v - b0) if v else -1 for v in row]
        print("  %-7s" % names[w], "sync %5d" % vals[1], "slots", " ".join("%5d" % v for v in vals[2:46] if v >= 0), "| end %5d %5d" % (vals[46], vals[47]))
# averages over all workgroups: per-slot completion time of the logic wave and the slowest emit wave
rel = d - d[:, :, 0].min(1)[:, None, None]
lg = rel[:, 0, 2:36].mean(0)
em = rel[:, 5:, 2:36].max(1).mean(0)
pr = rel[:, 1, 2:36].mean(0)
ap = rel[:, 2, 2:36].mean(0)
print('mean over workgroups: apply done time per slot :', ' '.join('%5d' % v for v in ap))
print('mean over workgroups: prep post time per slot  :', ' '.join('%5d' % v for v in pr))
print("mean over workgroups: logic post time per slot :", " ".join("%5d" % v for v in lg))
print("mean over workgroups: slowest emit done / slot :", " ".join("%5d" % v for v in em))
ph = d[:, 2, 38:43].mean(0)
print("apply wave phases (mean cycles per launch): wait %d  deal-loop %d  copy-loop %d  calls %d ; deals %.1f" % tuple(ph))
dp = d[:, 2, [36, 43, 44, 45, 37]].mean(0)
print("deal passes %.1f per launch; cycles: select %d  ring-read %d  compute+issue-writes %d  drain-writes %d" % tuple(dp))
print("mean sync stamp", rel[:, :, 1].mean(), "mean end", rel[:, :, 46].max(1).mean(), rel[:, :, 47].max(1).mean())
