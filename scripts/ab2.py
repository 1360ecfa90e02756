"""Same-box A/B of k_rollout_ws builds.  usage: python scripts/ab2.py name[=flags] ...
Each variant is the library (every csrc/*.hip) compiled with its -D flags (e.g. `base exp1=-DBRL_EXP=1`) into brl_amd/lib/variants/<name>.so
when built here (CPU container, `--build`), and timed on the GPU box in alternating subprocesses (`--run`): median of 5
repeats of 128 back-to-back launches between one event pair, 3 rotating output buffers.  `--check` also compares
every variant's Transition against the first variant's (bit-exact)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "brl_amd", "lib", "variants")
SRC = sorted(__import__("glob").glob(os.path.join(ROOT, "brl_amd", "csrc", "*.hip")))

BODY = r'''
import os, sys, json, hashlib
sys.path.insert(0, ROOT)
from brl_amd import _capi
_capi.LIB_PATH = LIB
import numpy as np, torch, ctypes as C
import brl_amd
from brl_amd.roll_out import alloc_transition
from bench import synthetic_lut
N, T = 8192, 32
keys, values = synthetic_lut(100000, 0)
env = brl_amd.BridgeBidding(lut=(keys, values))
NB = int(os.environ.get('NBUF', '3'))
trajs = [alloc_transition(T, N, env.device) for _ in range(NB)]
st = env.init(0, num_envs=N)
ptrs = []
for tr in trajs:
    p = _capi.TransitionPtrs()
    for f in _capi.TransitionPtrs._names:
        setattr(p, f, getattr(tr, f).data_ptr())
    ptrs.append(p)
lo = torch.empty((N, 480), dtype=torch.bool, device=env.device); lm = torch.empty((N, 38), dtype=torch.bool, device=env.device)
tc = torch.zeros(1, dtype=torch.int64, device=env.device)
s = torch.cuda.current_stream()
def launch(i):
    _capi.check(_capi.lib().brl_rollout_random(env._h, st.packed.data_ptr(), N, T, 1, (i * T) & 0xFFFFFFFF, 7600.0, C.byref(ptrs[i % NB]),
                                               lo.data_ptr(), lm.data_ptr(), tc.data_ptr(), s.cuda_stream))
launch(0); torch.cuda.synchronize()
h = hashlib.sha256()
for f in _capi.TransitionPtrs._names:
    h.update(getattr(trajs[0], f).cpu().numpy().tobytes())
h.update(st.packed.cpu().numpy().tobytes()); h.update(lo.cpu().numpy().tobytes())
for i in range(1, 10): launch(i)
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(s)
    for i in range(128): launch(10 + rep * 128 + i)
    e1.record(s); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 128 * 1e3)
iso = []
for i in range(40):   # isolated launches: one event pair each, host sync in between
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(s); launch(1000 + i); e1.record(s); torch.cuda.synchronize()
    iso.append(e0.elapsed_time(e1) * 1e3)
from brl_amd.gae import gae_scan
lv = torch.zeros(N, dtype=torch.float32, device=env.device)
mix = []
for rep in range(3):  # the bench's step: rollout + GAE alternating, 128 steps per event pair
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(s)
    for i in range(128):
        launch(2000 + rep * 128 + i); tb = trajs[i % NB]; gae_scan(env, tb.done, tb.value, tb.reward, lv, 1.0, 0.95)
    e1.record(s); torch.cuda.synchronize()
    mix.append(e0.elapsed_time(e1) / 128 * 1e3)
fus = []
if hasattr(_capi.lib(), "brl_rollout_random_gae"):  # the bench's step as ONE launch (rollout + calc_gae's scan)
    advs = [torch.empty((T, N), dtype=torch.float32, device=env.device) for _ in range(NB)]; tgts = [torch.empty_like(a) for a in advs]
    def launch_gae(i):
        _capi.check(_capi.lib().brl_rollout_random_gae(env._h, st.packed.data_ptr(), N, T, (i * T) & 0xFFFFFFFF, 7600.0, C.byref(ptrs[i % NB]),
                                                       lo.data_ptr(), lm.data_ptr(), tc.data_ptr(), lv.data_ptr(), 1.0, 0.95,
                                                       advs[i % NB].data_ptr(), tgts[i % NB].data_ptr(), s.cuda_stream))
    for i in range(10): launch_gae(3000 + i)
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(s)
        for i in range(128): launch_gae(3010 + rep * 128 + i)
        e1.record(s); torch.cuda.synchronize()
        fus.append(e0.elapsed_time(e1) / 128 * 1e3)
    h.update(advs[0].cpu().numpy().tobytes()); h.update(tgts[0].cpu().numpy().tobytes())
print(json.dumps({"fused": round(float(np.median(fus)), 2) if fus else None, "us": round(float(np.median(ts)), 2), "min": round(min(ts), 2), "iso": round(float(np.median(iso)), 2),
                  "step": round(float(np.median(mix)), 2), "sha": h.hexdigest()[:12]}))
'''

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    variants = [(a.split("=", 1)[0], a.split("=", 1)[1].split(",") if "=" in a else []) for a in args]
    if "--build" in sys.argv:
        os.makedirs(VDIR, exist_ok=True)
        for name, flags in variants:
            out = os.path.join(VDIR, name + ".so")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]
                                  + [f for f in flags if not f.startswith("env:")] + ["-o", out] + SRC, stderr=subprocess.DEVNULL)
            print("built", out)
    if "--run" in sys.argv:
        res = {n: [] for n, _ in variants}
        sha = {}
        extra = {}
        for rep in range(int(os.environ.get("REPS", "3"))):
            for name, flags in variants:
                code = f"ROOT={ROOT!r}\nLIB={os.path.join(VDIR, name + '.so')!r}\n" + BODY
                envv = dict(os.environ)
                envv.update(dict(f[4:].split("=", 1) for f in flags if f.startswith("env:")))  # e.g. env:BRL_ROLLOUT_FS=0
                r = subprocess.run(["timeout", "-k", "5", "90", sys.executable, "-c", code], capture_output=True, text=True, env=envv)
                try:
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    res[name].append(d["us"]); sha[name] = d["sha"]; extra.setdefault(name, []).append((d["iso"], d["step"], d.get("fused")))
                except Exception:
                    print(name, "FAILED", r.stderr[-800:])
        base = sha.get(variants[0][0])
        for name, _ in variants:
            v = sorted(res[name])
            print(f"{name:14s} median {v[len(v)//2] if v else None}  all {v}  isolated / step(rollout+gae) / one-launch step {extra.get(name)}  {'same-bytes' if sha.get(name) == base else 'DIFFERENT OUTPUT'}")

main()
