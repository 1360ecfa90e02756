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
