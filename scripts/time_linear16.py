"""brl_linear_act (csrc/mlp_infer.hpp) against the library GEMM: correctness vs a float64 product of the same 16-bit operands,
then interleaved timing rounds in ONE process (200 back-to-back launches between one event pair per round).
usage: python scripts/time_linear16.py [M] [--lib path/to/variant.so] [--skip-check] [--stamps]
(variants: `hipcc ... -DLIN16_EXP=1|2|4` = no DMA / no MFMA / no fragment reads inside the K loop; `-DLIN16_TIMING` for --stamps)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from brl_amd import _capi  # noqa: E402
from brl_amd.bridge_bidding import BridgeBidding, _stream  # noqa: E402


def main():
    argv = sys.argv[1:]
    if "--lib" in argv:
        _capi.LIB_PATH = os.path.abspath(argv[argv.index("--lib") + 1])
        del argv[argv.index("--lib"):argv.index("--lib") + 2]
    skip_check, stamps = "--skip-check" in argv, "--stamps" in argv
    argv = [a for a in argv if not a.startswith("--")]
    M = int(argv[0]) if argv else 8192
    dev = torch.device("cuda:0")
    env = BridgeBidding(device=dev)
    lib = _capi.lib()
    torch.manual_seed(0)

    def run(x, w, b, y, relu, fmt):
        _capi.check(lib.brl_linear_act(env._h, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), b.data_ptr() if b is not None else None,
                                     y.data_ptr(), y.stride(0), x.shape[0], w.shape[0], x.shape[1], int(relu), fmt, _stream()))

    # ---- correctness
    for dt, fmt in (() if skip_check else ((torch.bfloat16, 1), (torch.float16, 2))):
        for (m, n, k) in ((M, 1024, 1024), (M, 1024, 480), (300, 256, 64), (257, 128, 8), (1000, 1024, 200), (5, 128, 1024)):
            x = (torch.rand(m, k, device=dev) * 2 - 1).to(dt)
            if k == 480:
                x = (torch.rand(m, k, device=dev) < 0.2).to(dt)
            w = (torch.randn(n, k, device=dev) / k ** 0.5).to(dt)
            b = (torch.randn(n, device=dev) * 0.1).to(dt).float()
            y = torch.full((m, n), float("nan"), device=dev).to(dt)
            for relu in (1, 0):
                run(x, w, b, y, relu, fmt)
                torch.cuda.synchronize()
                ref = x.double() @ w.double().t() + b.double()
                if relu:
                    ref = ref.clamp_min(0)
                err = (y.double() - ref).abs()
                tol = ref.abs() * 2.0 ** (-8 if fmt == 1 else -11) + 1e-3   # half an ulp of the 16-bit output + accumulation slack
                bad = int((err > tol).sum())
                print(f"fmt {fmt} m {m} n {n} k {k} relu {relu}: max err {float(err.max()):.3e} (|ref| max {float(ref.abs().max()):.2f}) bad {bad}")
                assert bad == 0 and not torch.isnan(y.float()).any()
    if not skip_check:
        print("correct")
    if stamps:
        import ctypes
        dt = torch.bfloat16
        x = (torch.rand(M, 1024, device=dev) * 2 - 1).to(dt)
        w = (torch.randn(1024, 1024, device=dev) / 32).to(dt)
        bf = torch.zeros(1024, device=dev)
        y = torch.empty(M, 1024, device=dev, dtype=dt)
        nblk = (M + 255) // 256 * 8
        dbg = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
        hw = (torch.randn(39, 1024, device=dev) / 32).to(dt)
        parts = torch.empty(8, M, 40, device=dev)
        with_heads = "--heads" in sys.argv

        def one():
            if with_heads:
                _capi.check(lib.brl_linear_act_heads(env._h, x.data_ptr(), 1024, w.data_ptr(), 1024, bf.data_ptr(), y.data_ptr(), 1024, M, 1024,
                                                     1024, 1, 1, hw.data_ptr(), 1024, 39, parts.data_ptr(), 40, M * 40, _stream()))
            else:
                run(x, w, bf, y, 1, 1)

        for _ in range(50):
            one()
        lib.brl_lin16_set_dbg(ctypes.c_void_p(dbg.data_ptr()))
        one()
        torch.cuda.synchronize()
        lib.brl_lin16_set_dbg(ctypes.c_void_p(0))
        d = dbg.cpu().double()
        t0 = d[:, 0].min()
        print("shader-clock cycles, mean over workgroups (min .. max): start after the first workgroup's start | prologue "
              "(first chunk in registers) | K loop | epilogue")
        print(f"  start {float((d[:, 0] - t0).mean()):.0f} ({float((d[:, 0] - t0).min()):.0f} .. {float((d[:, 0] - t0).max()):.0f})")
        for a, b_, name in ((1, 0, "prologue"), (2, 1, "K loop"), (5, 2, "wait for the other waves"), (6, 5, "pack + LDS + barrier"),
                            (4, 6, "y stores issued"), (3, 4, "heads' share")):
            v = d[:, a] - d[:, b_]
            print(f"  {name:9s} {float(v.mean()):.0f} ({float(v.min()):.0f} .. {float(v.max()):.0f})")
        print(f"  whole launch: first start .. last end {float(d[:, 3].max() - t0):.0f}")
        return

    if "--heads" in sys.argv:   # the last hidden layer with the policy heads' share (brl_linear_act_heads) against the plain layer
        dt = torch.bfloat16
        x = (torch.rand(M, 1024, device=dev) * 2 - 1).to(dt)
        w = (torch.randn(1024, 1024, device=dev) / 32).to(dt)
        bf = torch.zeros(1024, device=dev)
        hw = (torch.randn(39, 1024, device=dev) / 32).to(dt)
        y = torch.empty(M, 1024, device=dev, dtype=dt)
        parts = torch.empty(8, M, 40, device=dev)

        def plain():
            run(x, w, bf, y, 1, 1)

        def heads(store):
            _capi.check(lib.brl_linear_act_heads(env._h, x.data_ptr(), 1024, w.data_ptr(), 1024, bf.data_ptr(), y.data_ptr() if store else None,
                                                 1024, M, 1024, 1024, 1, 1, hw.data_ptr(), 1024, 39, parts.data_ptr(), 40, M * 40, _stream()))

        fs = (("plain layer", plain), ("+ heads, y stored", lambda: heads(True)), ("+ heads, y not stored", lambda: heads(False)))
        for _, f in fs:
            for _ in range(20):
                f()
        torch.cuda.synchronize()
        res = {k: [] for k, _ in fs}
        for rnd in range(5):
            for name, f in fs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) * 1000 / 200)
        for name, v in res.items():
            v.sort()
            print(f"M {M} N 1024 K 1024  {name:22s} median {v[2]:.2f} us  min {v[0]:.2f} us")
        return

    # ---- timing: the four layers' shapes
    dt = torch.bfloat16
    for k in (1024, 480):
        x = (torch.rand(M, k, device=dev) * 2 - 1).to(dt)
        w = (torch.randn(1024, k, device=dev) / k ** 0.5).to(dt)
        wt = w.t().contiguous()
        b = (torch.randn(1024, device=dev) * 0.1).to(dt)
        bf = b.float()
        y = torch.empty(M, 1024, device=dev, dtype=dt)
        y2 = torch.empty(M, 1024, device=dev, dtype=dt)

        def ours():
            run(x, w, bf, y, 1, 1)

        def library():
            torch._addmm_activation(b, x, wt, use_gelu=False, out=y2)

        for f in (ours, library):
            for _ in range(20):
                f()
        torch.cuda.synchronize()
        print("max |ours - library|:", float((y.float() - y2.float()).abs().max()))
        res = {"ours": [], "library": []}
        for rnd in range(5):
            for name, f in (("ours", ours), ("library", library)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) * 1000 / 200)
        fl = 2.0 * M * 1024 * k
        for name, v in res.items():
            v.sort()
            print(f"M {M} N 1024 K {k}  {name:8s} median {v[2]:.2f} us  min {v[0]:.2f} us  = {fl / v[2] / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
