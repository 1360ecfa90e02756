"""Randomised soak of brl_linear_act / brl_linear_act_heads (csrc/mlp_infer.hpp): many shapes (M ragged, K any multiple of 8,
N any multiple of 128, strided operands), every result against the float64 product of the same 16-bit operands, and repeated
launches of the same problem compared bit for bit (a staging race would show as a run-to-run difference).
usage: python scripts/soak_linear16.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from brl_amd import _capi  # noqa: E402
from brl_amd.bridge_bidding import BridgeBidding, _stream  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    dev = torch.device("cuda:0")
    env = BridgeBidding(device=dev)
    L = _capi.lib()
    g = torch.Generator(device=dev).manual_seed(1234)
    rnd = torch.Generator().manual_seed(99)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rnd))

    t0, cases, launches = time.time(), 0, 0
    while time.time() - t0 < budget:
        fmt = ri(1, 2)
        dt = torch.bfloat16 if fmt == 1 else torch.float16
        big = ri(0, 3) == 0
        m = ri(1, 20000) if big else ri(1, 1500)
        n = 128 * ri(1, 8 if big else 12)
        k = 8 * ri(1, 160)
        ldx, ldw, ldy = k + 8 * ri(0, 3), k + 8 * ri(0, 3), n + 8 * ri(0, 3)
        x = ((torch.rand(m, ldx, device=dev, generator=g) * 2 - 1)).to(dt)
        w = (torch.randn(n, ldw, device=dev, generator=g) / k ** 0.5).to(dt)
        b = (torch.randn(n, device=dev, generator=g) * 0.1).to(dt).float()
        relu = ri(0, 1)
        heads = ri(0, 2) == 0
        nh = ri(1, 48)
        hw = (torch.randn(nh, n, device=dev, generator=g) / n ** 0.5).to(dt)
        hp_ld = ((nh + 3) // 4) * 4 + 4 * ri(0, 2)
        ys, ps = [], []
        for rep in range(3):
            y = torch.full((m, ldy), float("nan"), device=dev).to(dt)
            if heads:
                parts = torch.full((n // 128, m, hp_ld), float("nan"), device=dev)
                _capi.check(L.brl_linear_act_heads(env._h, x.data_ptr(), ldx, w.data_ptr(), ldw, b.data_ptr(), y.data_ptr(), ldy, m, n, k, relu,
                                                   fmt, hw.data_ptr(), n, nh, parts.data_ptr(), hp_ld, m * hp_ld, _stream()))
                ps.append(parts)
            else:
                _capi.check(L.brl_linear_act(env._h, x.data_ptr(), ldx, w.data_ptr(), ldw, b.data_ptr(), y.data_ptr(), ldy, m, n, k, relu, fmt,
                                             _stream()))
            ys.append(y)
            launches += 1
        torch.cuda.synchronize()
        what = (fmt, m, n, k, ldx, ldw, ldy, relu, heads, nh)
        for y in ys[1:]:
            assert torch.equal(y.view(torch.int16), ys[0].view(torch.int16)), ("run-to-run difference", what)
        for p in ps[1:]:
            assert torch.equal(p[:, :, :nh], ps[0][:, :, :nh]), ("run-to-run difference in the head parts", what)
        ref = x[:, :k].double() @ w[:, :k].double().t() + b.double()
        if relu:
            ref = ref.clamp_min(0)
        eps = 2.0 ** (-8 if fmt == 1 else -11)
        err = (ys[0][:, :n].double() - ref).abs()
        assert not torch.isnan(ys[0][:, :n].float()).any(), ("NaN left in y", what)
        assert int((err > ref.abs() * eps + 1e-3).sum()) == 0, ("y off", what, float(err.max()))
        if ldy > n:
            assert torch.isnan(ys[0][:, n:].float()).all(), ("wrote beyond column n", what)
        if heads:
            hd = ps[0][:, :, :nh].double().sum(0)
            want = ys[0][:, :n].double() @ hw.double().t()
            scale = float(want.abs().max()) + 1e-6
            assert float((hd - want).abs().max()) < 3e-4 * scale + 1e-4, ("head parts off", what, float((hd - want).abs().max()), scale)
        cases += 1
    print(f"soak ok: {cases} random problems, {launches} launches in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
