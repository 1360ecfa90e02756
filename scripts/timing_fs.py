"""Timing build (-DBRL_TIMING) of k_rollout_fs: per-wave length, emit waves' time spent waiting for the logic wave, and stamps
(cycles since wave start) at slots 0, 8, 16, 24, 32 of the logic wave and of the emit waves.  NBUF rotating output buffers."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libbrl_timing_fs.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-DBRL_TIMING"]
                      + os.environ.get("FLAGS", "").split() + ["-o", so] + sorted(__import__("glob").glob(os.path.join(ROOT, "brl_amd/csrc/*.hip"))), stderr=subprocess.DEVNULL)   # (every unit of the library)
from brl_amd import _capi
_capi.LIB_PATH = so
import numpy as np, torch, ctypes as C
import brl_amd
from brl_amd.roll_out import alloc_transition
from brl_amd.bridge_bidding import _stream
from bench import synthetic_lut
N, T, NW = 8192, 32, int(os.environ.get("NW", "13"))
keys, values = synthetic_lut(100000, 0)
env = brl_amd.BridgeBidding(lut=(keys, values))
NB = int(os.environ.get("NBUF", "3"))
trajs = [alloc_transition(T, N, env.device) for _ in range(NB)]
st = env.init(0, num_envs=N)
ps = []
for traj in trajs:
    p = _capi.TransitionPtrs()
    for f in _capi.TransitionPtrs._names:
        setattr(p, f, _capi.ptr(getattr(traj, f)))
    ps.append(p)
nblk = N // 32
dump = torch.zeros(nblk * NW * 2 + nblk * NW * 32, dtype=torch.int64, device=env.device)
lo = torch.empty((N, 480), dtype=torch.bool, device=env.device); lm = torch.empty((N, 38), dtype=torch.bool, device=env.device)
GAE = os.environ.get("GAE") == "1"   # brl_rollout_random_gae: the prep wave also scans the trajectory (calc_gae)
lv = torch.zeros(N, device=env.device); adv = torch.empty((T, N), device=env.device); tgt = torch.empty((T, N), device=env.device)
for i in range(4 * NB + 1):
    if GAE:
        _capi.check(_capi.lib().brl_rollout_random_gae(env._h, _capi.ptr(st.packed), N, T, i * T, 7600.0, C.byref(ps[i % NB]), _capi.ptr(lo), _capi.ptr(lm), _capi.ptr(dump), _capi.ptr(lv), 1.0, 0.95, _capi.ptr(adv), _capi.ptr(tgt), _stream()))
    else:
        _capi.check(_capi.lib().brl_rollout_random(env._h, _capi.ptr(st.packed), N, T, 1, i * T, 7600.0, C.byref(ps[i % NB]), _capi.ptr(lo), _capi.ptr(lm), _capi.ptr(dump), _stream()))
torch.cuda.synchronize()
full = dump.cpu().numpy()
d = full[:nblk * NW * 2].reshape(nblk, NW, 2)
tl = full[nblk * NW * 2:].reshape(nblk, NW, 16, 2)[..., 0]
names = ["logic", "loader/mask", "scorerA"] + [f"emit{g}" for g in range(8)] + ["scorerB", "prep"] + ["prologue-only"] * (NW - 13)
print("role          total(mean/p95)   wait(mean)   stamps (mean over workgroups, cycles/100)")
for w in range(NW):
    m = tl[:, w].mean(0)
    stamps = " ".join(f"{int(x) // 100:4d}" for x in m[:9] if x > 0) + "  | prologue: philox %d loads %d images %d barrier %d" % tuple(int(x) for x in m[9:13])
    print(f"{names[w]:12s} {d[:, w, 0].mean() / 100:6.0f} /{np.percentile(d[:, w, 0], 95) / 100:6.0f}   {d[:, w, 1].mean() / 100:6.0f}       {stamps}")
rt = full[nblk * NW * 2:].reshape(nblk, NW, 32)[:, :, 30]
print("shader clock: %.2f GHz (cycles / 100 MHz ticks of the slowest wave); launch = %.1f us by that wave's realtime" % (
    (d[..., 0].max(1) / np.maximum(rt.max(1), 1)).mean() * 0.1, rt.max(1).mean() / 100.0))
print("slowest wave per workgroup: mean %.0f p95 %.0f max %.0f" % tuple(x / 100 for x in (d[..., 0].max(1).mean(), np.percentile(d[..., 0].max(1), 95), d[..., 0].max())))
