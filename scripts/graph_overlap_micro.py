"""Do two branches of a captured hipGraph run side by side on this ROCm?  A spin kernel (8 workgroups, 100 us) on a side stream beside
five 1024^3 fp32 GEMMs (~19 us each) on the main stream: eager two-stream launch vs the same captured into one graph.
Serial = ~200 us, overlapped = ~100 us.  (scripts/micro/libspin.so: scripts/overlap_probe.py builds it.)"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spin = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libspin.so"))
spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
dev = torch.device("cuda", 0)
a, b, c = (torch.randn(1024, 1024, device=dev) for _ in range(3))
side = torch.cuda.Stream()


def body(with_spin, blocks=8, usec=100.0):
    cur = torch.cuda.current_stream()
    if with_spin:
        side.wait_stream(cur)
        spin.spin_launch(blocks, usec, ctypes.c_void_p(side.cuda_stream))
    for _ in range(5):
        torch.mm(a, b, out=c)
    if with_spin:
        cur.wait_stream(side)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


out = []
out.append(f"eager, GEMMs only           : {timed(lambda: body(False)):7.1f} us")
out.append(f"eager, spin beside the GEMMs: {timed(lambda: body(True)):7.1f} us")
for blocks in (8, 64):
    for mode in ("global", "thread_local", "relaxed"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=mode):
            body(True, blocks)
        out.append(f"graph ({mode:12s}, {blocks:2d} spin workgroups): {timed(g.replay):7.1f} us")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body(False)
out.append(f"graph, GEMMs only           : {timed(g.replay):7.1f} us")
print("\n".join(out))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(out) + "\n")
