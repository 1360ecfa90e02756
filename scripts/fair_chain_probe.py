"""brl_fair_chain alone: us per launch at minibatch 1024 for each library given (experiment builds: -DFAIR_EXP=n, see
brl_amd/csrc/fair_chain.hpp), 200 back-to-back launches between one event pair, median of 5.
    python scripts/fair_chain_probe.py [--build name=-Dflag ...]      (CPU container: builds brl_amd/lib/variants/<name>.so)
    python scripts/fair_chain_probe.py name ...                         (GPU box: times them, one subprocess each)"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "brl_amd", "lib", "variants")

BODY = r'''
import os, sys, ctypes as C
sys.path.insert(0, ROOT)
from brl_amd import _capi
_capi.LIB_PATH = LIB
import numpy as np, torch
dev = torch.device("cuda", 0)
B, H = 1024, 200
g = torch.Generator(device=dev).manual_seed(0)
f = lambda *s: torch.randn(s, device=dev, generator=g) * 0.05
net, wk = _capi.FairNet(), _capi.FairWork()
keep = []
for l in range(11):
    w = f(H, 480 if l == 0 else 680 if l == 6 else H); b = f(H); keep += [w, b]
    net.w[l], net.b[l] = w.data_ptr(), b.data_ptr()
wh, bh = f(39, H), f(39); keep += [wh, bh]
net.head_w, net.head_b = wh.data_ptr(), bh.data_ptr()
nwg = B // 16
bufs = dict(inp=(9, B, H), dzs=(9, B, H), gates=(4, B, H), cat6=(B, 680), x4=(B, H), dz0=(B, H), dz6=(B, H), dheads=(B, 40),
            tiles=(11 * nwg * H + nwg * 39,), partials=(nwg, 8), gram_partials=(nwg, 1444))
for k, shp in bufs.items():
    t = torch.zeros(shp, device=dev); keep.append(t); setattr(wk, k, t.data_ptr())
x0 = (torch.rand((B, 480), device=dev, generator=g) < 0.1).float()
mask = (torch.rand((B, 38), device=dev, generator=g) < 0.5).to(torch.uint8); mask[:, 0] = 1
action = torch.zeros(B, dtype=torch.int32, device=dev)
ov, olp, adv, tgt = f(B), -torch.rand(B, device=dev, generator=g) - 0.5, f(B), f(B)
s = torch.cuda.current_stream()
L = _capi.lib()
def launch():
    _capi.check(L.brl_fair_chain(0, net, x0.data_ptr(), mask.data_ptr(), action.data_ptr(), ov.data_ptr(), olp.data_ptr(), adv.data_ptr(),
                                 tgt.data_ptr(), B, 0.2, 0.5, 0.01, 1, 1, 0, 0, wk, s.cuda_stream))
launch(); torch.cuda.synchronize()
# forward check against torch in float64: h0 = relu(x0 W0^T + b0) (inp[0]) and h1 (inp[1])
W0, b0, W1, b1 = keep[0].double(), keep[1].double(), keep[2].double(), keep[3].double()
h0 = torch.relu(x0.double() @ W0.t() + b0); h1 = torch.relu(h0 @ W1.t() + b1)
inp_t = keep[24]
print(f"{NAME}: |inp[0] - h0| = {float((inp_t[0].double() - h0).abs().max()):.2e}, |inp[1] - h1| = {float((inp_t[1].double() - h1).abs().max()):.2e}")
if os.environ.get("FAIR_DUMP"):     # every output of one launch -> <dir>/<name>.npz (compare two builds on the same inputs)
    np.savez(os.path.join(os.environ["FAIR_DUMP"], NAME + ".npz"), **{k: t.cpu().numpy() for k, t in zip(bufs, keep[24:])})
for _ in range(20): launch()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(s)
    for _ in range(200): launch()
    e1.record(s); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 200 * 1e3)
print(f"{NAME:16s} {np.median(ts):8.2f} us per launch (min {min(ts):.2f})")
if hasattr(L, "brl_fair_set_dbg"):     # a -DFAIR_TIMING build: shader-clock stamps of thread 0 behind every phase, per workgroup
    dbg = torch.zeros((nwg, 64), dtype=torch.int64, device=dev)
    L.brl_fair_set_dbg.argtypes = [C.c_void_p]
    _capi.check(L.brl_fair_set_dbg(dbg.data_ptr()))
    launch(); torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    n = int((d[0] != 0).sum())
    names = ["start", "prologue", "L0"] + [f"L{l}" for l in (1, 2, 3, 4, 5)] + ["L6", "L7", "L8", "L9", "L10", "heads", "loss", "dx heads"] \
        + ["b10", "b9", "b8", "b7", "b6", "b5", "b4", "b3", "b2", "b1"]
    for wg in (0, 1, nwg // 2, nwg - 1):
        seg = np.diff(d[wg, :n])
        print(f"  workgroup {wg:3d}: total {int(d[wg, n - 1] - d[wg, 0])} ticks; " + "  ".join(f"{names[i + 1] if i + 1 < len(names) else i}:{int(v)}" for i, v in enumerate(seg)))
    print("  (s_memtime ticks: 100 MHz on gfx950 -> 10 ns each)")
'''


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--build" in sys.argv:
        os.makedirs(VDIR, exist_ok=True)
        src = sorted(glob.glob(os.path.join(ROOT, "brl_amd", "csrc", "*.hip")))
        for a in args:
            name, flags = (a.split("=", 1) + [""])[:2]
            out = os.path.join(VDIR, name + ".so")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                                   "-I" + os.path.join(ROOT, "include")] + [x for x in flags.split(",") if x] + ["-o", out] + src,
                                  stderr=subprocess.DEVNULL)
            print("built", out)
        return
    for name in args:
        lib = os.path.join(VDIR, name + ".so") if name != "product" else os.path.join(ROOT, "brl_amd", "lib", "libbrl_hip.so")
        code = f"ROOT={ROOT!r}\nLIB={lib!r}\nNAME={name!r}\n" + BODY
        r = subprocess.run(["timeout", "-k", "5", "120", sys.executable, "-c", code], capture_output=True, text=True)
        print(r.stdout.strip() if r.stdout.strip() else (name + " FAILED " + r.stderr[-600:]))


main()
