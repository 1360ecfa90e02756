"""Functional proof of scripts/micro/peer_adam.hip (the gradient collective fused into the optimizer launches: peers' buffers mapped with
hipIpcOpenMemHandle, three flag words per peer instead of collective kernels) — WORLD processes sharing the ONE GPU of this box, three
optimizer steps with fresh gradients each:
    python scripts/peer_adam_probe.py [world=2] [n=1048576] [out file]
Checks: no wait timed out; every rank ends with BIT-IDENTICAL parameters (each rank applied Adam to its own slice only and received the
other slices through the peers' stores); they equal a float64 restatement of mean-gradient -> clip_by_global_norm -> Adam (the arithmetic of
csrc/adam_role.hpp) to fp32 rounding; the moments exist on the owner's slice only.  No torch: ctypes on libamdhip64 + the prototype's .so."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "scripts", "micro", "libpeer_adam.so")
SRC = os.path.join(ROOT, "scripts", "micro", "peer_adam.hip")
LR, B1, B2, EPS, MAXN, STEPS = 1e-3, 0.9, 0.999, 1e-8, 0.5, 3


class Handle(C.Structure):
    _fields_ = [("b", C.c_ubyte * 64)]


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, SRC])


def grad_of(rank, step, n):
    return (np.random.default_rng(1000 * step + rank).standard_normal(n) * (0.01 * (1 + rank))).astype(np.float32)


def worker(rank, world, n, conns, result_q):
    hip = C.CDLL("libamdhip64.so")
    lib = C.CDLL(SO)

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"rank {rank}: {what} -> {rc}")
    chk(hip.hipSetDevice(0), "hipSetDevice")
    hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    ctl_bytes = lib.peer_ctl_bytes()

    def malloc(nbytes):
        p = C.c_void_p()
        chk(hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)), "hipMalloc")
        chk(hip.hipMemset(p, 0, C.c_size_t(nbytes)), "hipMemset")
        return p
    G, P, ctl = malloc(4 * n), malloc(4 * n), malloc(ctl_bytes)
    gred, m, v, norm = malloc(4 * n), malloc(4 * n), malloc(4 * n), malloc(4)
    p0 = (np.random.default_rng(7).standard_normal(n) * 0.1).astype(np.float32)
    chk(hip.hipMemcpy(P, p0.ctypes.data_as(C.c_void_p), C.c_size_t(4 * n), 1), "H2D")
    chk(hip.hipDeviceSynchronize(), "sync")
    mine = []
    for ptr in (G, P, ctl):
        h = Handle()
        chk(hip.hipIpcGetMemHandle(C.byref(h), ptr), "hipIpcGetMemHandle")
        mine.append(bytes(bytearray(h.b)))
    # all-to-all of the handles through the parent
    conns[rank].send(mine)
    if not conns[rank].poll(90):
        raise RuntimeError("no answer from the parent")
    everyone = conns[rank].recv()                    # [rank][3] handle bytes
    tabs = [(C.c_void_p * world)() for _ in range(3)]
    for q in range(world):
        for k, own in enumerate((G, P, ctl)):
            if q == rank:
                tabs[k][q] = own
            else:
                h = Handle()
                C.memmove(C.byref(h), everyone[q][k], 64)
                p = C.c_void_p()
                chk(hip.hipIpcOpenMemHandle(C.byref(p), h, 1), "hipIpcOpenMemHandle")   # hipIpcMemLazyEnablePeerAccess
                tabs[k][q] = p
    conns[rank].send("mapped")
    conns[rank].recv()                               # everybody has mapped everybody
    lib.peer_adam_step.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_ulonglong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    norms = []
    for step in range(1, STEPS + 1):
        g = grad_of(rank, step, n)
        chk(hip.hipMemcpy(G, g.ctypes.data_as(C.c_void_p), C.c_size_t(4 * n), 1), "H2D grad")
        chk(lib.peer_adam_step(rank, world, tabs[0], tabs[1], tabs[2], gred, m, v, n, step, LR, B1, B2, EPS, MAXN, norm, None), "peer_adam_step")
        chk(hip.hipDeviceSynchronize(), "sync")      # (its last launch waited for every peer's parameter flag: nobody reads my G any more)
        nb = np.zeros(1, np.float32)
        hip.hipMemcpy(nb.ctypes.data_as(C.c_void_p), norm, C.c_size_t(4), 2)
        norms.append(float(nb[0]))
    out_p, out_m = np.empty(n, np.float32), np.empty(n, np.float32)
    hip.hipMemcpy(out_p.ctypes.data_as(C.c_void_p), P, C.c_size_t(4 * n), 2)
    hip.hipMemcpy(out_m.ctypes.data_as(C.c_void_p), m, C.c_size_t(4 * n), 2)
    cb = np.zeros(ctl_bytes // 4, np.uint32)
    hip.hipMemcpy(cb.ctypes.data_as(C.c_void_p), ctl, C.c_size_t(ctl_bytes), 2)
    err = int(cb[3 * 8 * 2])                          # the word behind three arrays of 8 64-bit flags
    conns[rank].send("done")
    conns[rank].recv()                               # nobody unmaps while a peer still reads
    result_q.put((rank, out_p, out_m, norms, err))


def reference(world, n):
    p = (np.random.default_rng(7).standard_normal(n) * 0.1).astype(np.float32).astype(np.float64)
    m, v = np.zeros(n), np.zeros(n)
    norms = []
    for step in range(1, STEPS + 1):
        g = np.zeros(n, np.float32)
        for q in range(world):                       # (the kernel's sum: fp32, rank order)
            g = g + grad_of(q, step, n)
        g = g.astype(np.float64) / world
        norm = np.sqrt((g * g).sum())
        norms.append(norm)
        g = g * min(MAXN / (norm + 1e-6), 1.0)
        m = m + (g - m) * (1 - B1)
        v = v * B2 + g * g * (1 - B2)
        p = p - (LR / (1 - B1 ** step)) * (m / (np.sqrt(v) / np.sqrt(1 - B2 ** step) + EPS))
    return p, m, norms


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    assert n % (4 * world) == 0
    build()
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, n, [c[1] for c in pipes], q)) for r in range(world)]
    for p in procs:
        p.start()
    par = [c[0] for c in pipes]

    def gather_scatter(payload_of_all):
        got = []
        for c in par:
            if not c.poll(90):           # a rank died or hangs: never wait for ever on a GPU box
                for p in procs:
                    p.kill()
                print("FAIL: a rank did not answer within 90 s", flush=True)
                sys.exit(2)
            got.append(c.recv())
        for c in par:
            c.send(payload_of_all(got))
    gather_scatter(lambda got: got)                  # the handles
    gather_scatter(lambda got: "go")                 # mapped
    gather_scatter(lambda got: "bye")                # done
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    lines = []
    ok = all(r[4] == 0 for r in res)
    lines.append(f"world {world} on one GPU, n {n}, {STEPS} steps: wait time-outs {[r[4] for r in res]}")
    same = all(np.array_equal(res[0][1], r[1]) for r in res[1:])
    lines.append(f"parameters bit-identical on every rank: {same}")
    want_p, want_m, want_norms = reference(world, n)
    dp = float(np.abs(res[0][1].astype(np.float64) - want_p).max())
    lines.append(f"max |p - float64 restatement| {dp:.3e} (lr {LR}: {dp / LR:.2e} of a full step); norms {res[0][3]} vs {[round(x, 6) for x in want_norms]}")
    ln = n // world
    own = all(float(np.abs(res[r][2][r * ln:(r + 1) * ln].astype(np.float64) - want_m[r * ln:(r + 1) * ln]).max()) < 1e-6 for r in range(world))
    others_zero = all(not res[r][2][((r + 1) % world) * ln:((r + 1) % world + 1) * ln].any() for r in range(world)) if world > 1 else True
    lines.append(f"first moments: the owner's slice matches: {own}; the other ranks' slices untouched: {others_zero}")
    norms_ok = all(abs(a - b) < 1e-4 * b for a, b in zip(res[0][3], want_norms)) and all(r[3] == res[0][3] for r in res)
    ok = ok and same and dp < 2e-2 * LR and own and others_zero and norms_ok
    lines.append("PASS" if ok else "FAIL")
    print("\n".join(lines), flush=True)
    if out_path:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        open(out_path, "w").write("\n".join(lines) + "\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
