"""Timing of the large-batch inference layer: brl_linear_x3p (operands pre-split into bf16 planes; csrc/mlp_linear_x3p.hpp) beside brl_mlp_gemm_x3
(the split in registers) and torch's fp32 GEMM, at the policy rollout's shapes — hipEvents around back-to-back launches.
    python scripts/x3p_layer_probe.py [iters=100]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brl_amd import _capi   # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / iters)
    return out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    L = _capi.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K, npx in ((8192, 1024, 1024, 3), (8192, 1024, 480, 1), (8192, 1024, 480, 3), (10000, 1024, 1024, 3)):
        x = (torch.rand((M, K), device="cuda", generator=g) < 0.1).float() if npx == 1 else torch.rand((M, K), device="cuda", generator=g) * 2 - 1
        w = (torch.rand((N, K), device="cuda", generator=g) * 2 - 1) * 0.05
        b = torch.rand(N, device="cuda", generator=g) - 0.5
        wp = torch.empty((3, N, K), dtype=torch.bfloat16, device="cuda")
        _capi.check(L.brl_split_planes(0, w.data_ptr(), N * K, wp.data_ptr(), N * K, s))
        if npx == 3:
            xp = torch.empty((3, M, K), dtype=torch.bfloat16, device="cuda")
            _capi.check(L.brl_split_planes(0, x.data_ptr(), M * K, xp.data_ptr(), M * K, s))
        else:
            xp = x.to(torch.bfloat16)
        y = torch.empty((M, N), device="cuda")
        yp = torch.empty((3, M, N), dtype=torch.bfloat16, device="cuda")
        sx = M * K if npx == 3 else 0
        t_pl = timed(lambda: _capi.check(L.brl_linear_x3p(0, xp.data_ptr(), npx, K, sx, wp.data_ptr(), K, N * K, b.data_ptr(), 1, None, 0,
                                                          yp.data_ptr(), N, M * N, M, N, K, s)), iters)
        t_f32 = timed(lambda: _capi.check(L.brl_linear_x3p(0, xp.data_ptr(), npx, K, sx, wp.data_ptr(), K, N * K, b.data_ptr(), 1, y.data_ptr(), N,
                                                           None, 0, 0, M, N, K, s)), iters)
        t_x3 = timed(lambda: _capi.check(L.brl_mlp_gemm_x3(0, 0, 1, x.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), N, M, N, K, 0, b.data_ptr(),
                                                           None, 0, None, None, 0, s)), iters)
        wt = w.t().contiguous()
        t_lib = timed(lambda: torch._addmm_activation(b, x, wt, use_gelu=False, out=y), iters)
        flop = 2.0 * M * N * K
        fmt = lambda ts: "  ".join(f"{t:7.2f}" for t in ts)   # noqa: E731
        print(f"{M} x {N} x {K}, x as {npx} plane(s): us per launch (three rounds of {iters})")
        print(f"    brl_linear_x3p -> planes   {fmt(t_pl)}    ({flop / min(t_pl) / 1e6:.0f} TFLOP/s fp32-equivalent)")
        print(f"    brl_linear_x3p -> fp32     {fmt(t_f32)}")
        print(f"    brl_mlp_gemm_x3            {fmt(t_x3)}")
        print(f"    torch fp32 GEMM + bias + ReLU {fmt(t_lib)}", flush=True)


if __name__ == "__main__":
    main()
