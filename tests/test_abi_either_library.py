"""the same C symbols are exported by the CPU oracle (oracle/brl_shim.c) and by libbrl_hip.so, so ONE parity
scenario written against include/brl_hip.h (tests/abi_scenario.py) runs on either.  CPU: the shim exports every declared
symbol and reproduces the oracle's own Python binding; GPU: the product library gives the same bytes as the shim."""
import ctypes
import os

import numpy as np
import pytest

from tests import abi_scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    import oracle
    oracle.build()
    from oracle.binding import shim_path
    return ctypes.CDLL(shim_path())


def test_shim_exports_every_symbol_of_the_header(shim):
    from tests.test_capi_cpu import header_symbols
    for s in header_symbols():
        assert hasattr(shim, s), s
    shim.brl_last_error.restype = ctypes.c_char_p
    assert shim.brl_ppo_stats(0, None, 0, None, ctypes.c_float(0), ctypes.c_float(0), None, None) == -1
    assert b"oracle shim" in shim.brl_last_error()


def test_shim_scenario_matches_the_oracle_binding(shim, dds, oracle):
    """the scenario through the brl_* symbols == the same calls through oracle/binding.py (orc_* symbols)"""
    n, steps, seed = 257, 60, 20240611
    got = abi_scenario.run(shim, abi_scenario.HostMem(), dds["keys"], dds["values"], n, steps, seed)
    ref = oracle.init_random(n, seed=seed, env_offset=1000)
    assert np.array_equal(got["obs0"], ref["observation"]) and np.array_equal(got["mask0"], ref["legal_action_mask"])
    rng = np.random.default_rng(7)
    m = ref["legal_action_mask"].copy()
    for t in range(steps):
        act = (rng.random(m.shape) * m).argmax(1).astype(np.int32)
        act[rng.random(n) < 0.4] = 0
        oracle.step(ref, act, autoreset=True, seed=seed, env_offset=1000)
        m = ref["legal_action_mask"].copy()
        assert np.array_equal(got["step_obs"][t], ref["observation"]) and np.array_equal(got["step_rewards"][t], ref["rewards"])
        assert np.array_equal(got["step_terminated"][t], ref["terminated"].astype(np.uint8))
    assert np.array_equal(got["field_last_bid"], ref["last_bid"]) and np.array_equal(got["field_board_ctr"], ref["board_ctr"])
    want = oracle.rollout_random(ref, 12, seed=seed, env_offset=1000, draw_base=5)
    for name in ("obs", "legal_action_mask", "action", "reward", "done", "log_prob"):
        assert np.array_equal(got["rollout_" + name], want[name]), name
    assert int(got["rollout_count"][0]) == want["terminated_count"] and got["dup_iterations"] < 399
    assert np.abs(got["dup_cum_imp"]).max() <= 24 and (got["dup_A_terminated"] == 1).all() and (got["dup_B_terminated"] == 1).all()


@pytest.mark.gpu
def test_product_library_matches_the_oracle_on_the_same_abi_scenario(shim, dds):
    from brl_amd import _capi
    hip = ctypes.CDLL(_capi.LIB_PATH)
    want = abi_scenario.run(shim, abi_scenario.HostMem(), dds["keys"], dds["values"])
    got = abi_scenario.run(hip, abi_scenario.DeviceMem(), dds["keys"], dds["values"])
    assert set(got) == set(want)
    for k in sorted(want):
        g, w = got[k], want[k]
        if w.dtype in (np.uint32, np.uint64):
            g = g.view(w.dtype) if g.dtype != w.dtype else g
        assert g.shape == w.shape and np.array_equal(g, w), k


def test_shard_adam_on_the_shim_sharded_equals_replicated_equals_torch(shim):
    """brl_adam_shard_norm / brl_adam_shard_apply (the multi-rank step's clip + Adam, include/brl_hip.h) through the CPU shim, host
    pointers: three ranks' sharded sweeps (own partials, "all-gather" = one shared array, own slices) == one replicated sweep ==
    torch.optim.Adam + clip_grad_norm_; the counters advance once per step."""
    import torch
    from brl_amd._capi import ShardGeom
    f32, vp, i32, i64 = ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    shim.brl_adam_shard_norm.argtypes = [i32, vp, ctypes.POINTER(ShardGeom), i32, i32, f32, vp, vp, vp, vp]
    shim.brl_adam_shard_apply.argtypes = [i32, vp, vp, vp, vp, ctypes.POINTER(ShardGeom), i32, i32, vp, vp, f32, vp, f32, f32, f32, f32, f32, vp,
                                          vp, i64, vp]
    W, lens = 3, [64, 20, 8]                         # three buckets, slices of 64 / 20 / 8 floats
    geom = ShardGeom()
    geom.nbuckets, geom.world, geom.nsub = 3, W, 2
    off = 0
    for b, ln in enumerate(lens):
        geom.off[b], geom.len[b] = off, ln
        off += W * ln
    n = off
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(n).astype(np.float32)
    grads = [(rng.standard_normal(n) * (1e-3 if it == 1 else 1.0)).astype(np.float32) for it in range(3)]
    ptr = lambda a: a.ctypes.data_as(vp)   # noqa: E731

    def run(sharded):
        p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        step, idx, norm = np.zeros(1, np.float32), np.zeros(1, np.int32), np.zeros(1, np.float32)
        for g in grads:
            gs = (g * W).astype(np.float32)
            part = np.full(W * 3 * 2, np.nan, np.float32)
            for lo, hi in ([(r, r + 1) for r in range(W)] if sharded else [(0, W)]):
                st = step.copy() if sharded else step            # (every rank has its own counter; they agree)
                assert shim.brl_adam_shard_norm(0, ptr(gs), ctypes.byref(geom), lo, hi, f32(1.0 / W), ptr(part), ptr(st),
                                                ptr(idx) if lo == 0 else None, None) == 0
            if sharded:
                step[:] = st
            assert not np.isnan(part).any()
            for lo, hi in ([(r, r + 1) for r in range(W)] if sharded else [(0, W)]):
                assert shim.brl_adam_shard_apply(0, ptr(p), ptr(gs), ptr(m), ptr(v), ctypes.byref(geom), lo, hi, ptr(part), ptr(step),
                                                 f32(1e-3), None, f32(0.9), f32(0.999), f32(1e-5), f32(0.5), f32(1.0 / W), ptr(norm),
                                                 None, 0, None) == 0
        assert step[0] == 3.0 and idx[0] == 3
        return p, m, v, float(norm[0])
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([ref], lr=1e-3, eps=1e-5)
    for g in grads:
        ref.grad = torch.from_numpy(g.copy())
        want_norm = float(torch.nn.utils.clip_grad_norm_([ref], 0.5))
        opt.step()
    assert abs(a[3] - want_norm) < 1e-5 * want_norm
    assert np.allclose(a[0], ref.detach().numpy(), atol=2e-6)
    assert shim.brl_adam_shard_norm(0, ptr(p0), ctypes.byref(geom), 2, 2, f32(1.0), ptr(p0), ptr(p0), None, None) == -1   # empty rank range


def test_mlp_gemm_group_on_the_shim(shim):
    """brl_mlp_gemm_group through the shim (the plain definition, product by product) against numpy in float64: the FAIR step's
    TN products, including the heads' 39 rows out of a [K, 40] array (lda = m rounded up to 4)."""
    rng = np.random.default_rng(5)
    shapes = [(8, 12, 20, 8), (39, 16, 24, 40)]        # (m, n, k, lda)
    As = [np.ascontiguousarray(rng.standard_normal((k, lda)).astype(np.float32)) for m, n, k, lda in shapes]
    Bs = [np.ascontiguousarray(rng.standard_normal((k, n)).astype(np.float32)) for m, n, k, lda in shapes]
    Cs = [np.full((m, n), np.nan, np.float32) for m, n, k, lda in shapes]
    cnt = len(shapes)
    vp, i64 = ctypes.c_void_p * cnt, ctypes.c_int64 * cnt
    shim.brl_mlp_gemm_group.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 10
    rc = shim.brl_mlp_gemm_group(0, 2, cnt, vp(*[a.ctypes.data for a in As]), i64(*[s[3] for s in shapes]),
                                 vp(*[b.ctypes.data for b in Bs]), i64(*[s[1] for s in shapes]), vp(*[c.ctypes.data for c in Cs]),
                                 i64(*[s[1] for s in shapes]), i64(*[s[0] for s in shapes]), i64(*[s[1] for s in shapes]),
                                 i64(*[s[2] for s in shapes]), None)
    assert rc == 0
    for (m, n, k, lda), a, b, c in zip(shapes, As, Bs, Cs):
        assert np.allclose(c, a[:, :m].astype(np.float64).T @ b.astype(np.float64), atol=1e-5)
