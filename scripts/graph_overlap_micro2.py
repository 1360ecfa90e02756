"""How much does each GEMM kernel of the step lose when a few CUs are held beside it?  Five 1024^3 fp32 products in a captured graph, alone
and beside a 40 us spin kernel (shorter than the chain: the GEMM branch stays the critical path) of 8 / 32 / 128 / 256 workgroups on a second branch: the library's kernel (torch.mm) and brl_mlp_gemm with 64 x 32
tiles (512 workgroups, two per CU) and 64 x 64 tiles (256 workgroups, one per CU).  usage: python scripts/graph_overlap_micro2.py [out]"""
import ctypes
import os
import subprocess
import sys
import time

BODY = r'''
import ctypes, os, sys, time, torch
ROOT = os.environ["ROOT"]; sys.path.insert(0, ROOT)
from brl_amd import _capi
spin = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libspin.so"))
spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
dev = torch.device("cuda", 0)
a, b, c = (torch.randn(1024, 1024, device=dev) for _ in range(3))
side = torch.cuda.Stream()
kind = os.environ["KIND"]
L = _capi.lib()
def gemm():
    if kind == "library":
        torch.mm(a, b.t(), out=c)
    else:
        _capi.check(L.brl_mlp_gemm(0, 0, 0, a.data_ptr(), 1024, b.data_ptr(), 1024, c.data_ptr(), 1024, 1024, 1024, 1024, 0, None, None, 0, None, None,
                                   torch.cuda.current_stream().cuda_stream))
def body(blocks):
    cur = torch.cuda.current_stream()
    if blocks:
        side.wait_stream(cur)
        spin.spin_launch(blocks, 40.0, ctypes.c_void_p(side.cuda_stream))
    for _ in range(5):
        gemm()
    if blocks:
        cur.wait_stream(side)
res = []
for blocks in (0, 8, 32, 128, 256):
    for _ in range(3):
        body(blocks)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(blocks)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 50 * 1e6)
print("RESULT %s: alone %.1f us; beside a 40 us spin kernel of 8 / 32 / 128 / 256 workgroups: %.1f / %.1f / %.1f / %.1f us" % (os.environ["LABEL"], *res))
'''
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
for label, kind, tile in (("library (torch.mm)", "library", None), ("brl_mlp_gemm 64 x 32 tiles", "own", "32"), ("brl_mlp_gemm 64 x 64 tiles", "own", "64")):
    env = dict(os.environ, ROOT=ROOT, KIND=kind, LABEL=label)
    if tile:
        env["BRL_GEMM_TILE_N"] = tile
    r = subprocess.run([sys.executable, "-c", BODY], env=env, capture_output=True, text=True, timeout=200)
    line = next((l for l in r.stdout.splitlines() if l.startswith("RESULT")), "FAILED " + r.stderr[-300:])
    print(line, flush=True)
    out.append(line)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(out) + "\n")
