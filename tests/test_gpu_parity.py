"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.
Bit-exact for every integer / byte / index field; float fields are exact too unless a
tolerance is written in the test."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import synthetic_lut
from tests.gpu_util import assert_state_equal, random_legal_actions, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(dds):
    import brl_amd
    return brl_amd.BridgeBidding(lut=(dds["keys"], dds["values"]))


def make_env(dds, k, ws=None, lut=None):
    """k: tables per wave of the per-step kernels; ws: "0" for the K-tables-per-wave fused rollout
    (k_rollout_random<K>), "ws" for the barrier-synchronised wave-specialised kernel (k_rollout_ws) on every shape,
    None for the library default (the flag-synchronised k_rollout_fs where it applies: substeps 1, T <= 40, n % 32 == 0;
    k_rollout_ws otherwise)."""
    import brl_amd
    new = {"BRL_TABLES_PER_WAVE": str(k), "BRL_ROLLOUT_WS": "0" if ws == "0" else None,
           "BRL_ROLLOUT_FS": "0" if ws == "ws" else None}
    old = {key: os.environ.get(key) for key in new}
    for key, v in new.items():
        if v is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = v
    try:
        return brl_amd.BridgeBidding(lut=lut if lut is not None else (dds["keys"], dds["values"]))
    finally:
        for key, v in old.items():
            if v is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = v


def test_extension_is_the_in_tree_hip_library():
    from brl_amd import _capi
    assert os.path.exists(_capi.LIB_PATH)
    assert _capi.lib().brl_version() >= 1


@pytest.mark.parametrize("n", [1, 3, 17, 1000, 4096])
def test_init_random_matches_oracle(env, oracle, n):
    st = env.init(1234, num_envs=n)
    ref = oracle.init_random(n, seed=1234)
    assert_state_equal(st, ref, where=f"init n={n}")


def test_init_random_env_offset(dds, oracle):
    import brl_amd
    e = brl_amd.BridgeBidding(lut=(dds["keys"], dds["values"]), env_offset=5000)
    st = e.init(77, num_envs=64)
    ref = oracle.init_random(64, seed=77, env_offset=5000)
    assert_state_equal(st, ref, where="env_offset")


@pytest.mark.parametrize("k", [1, 2, 4, 8])
def test_step_random_auctions_no_reset(dds, oracle, k):
    env = make_env(dds, k)
    n = 1531  # ragged on purpose
    rng = np.random.default_rng(k)
    st = env.init(9, num_envs=n)
    ref = oracle.init_random(n, seed=9)
    for step in range(70):
        act = random_legal_actions(rng, ref["legal_action_mask"])
        if step % 3 == 0:  # bias towards passes so that auctions end at all lengths
            act[rng.random(n) < 0.5] = 0
        st = env.step(st, torch.from_numpy(act))
        oracle.step(ref, act)
        if step % 7 == 0 or step > 60:
            assert_state_equal(st, ref, where=f"K={k} step {step}")
    assert ref["terminated"].mean() > 0.9


@pytest.mark.parametrize("k", [1, 4, 8])
def test_step_autoreset(dds, oracle, k):
    env = make_env(dds, k)
    n = 777
    rng = np.random.default_rng(100 + k)
    st = env.init(31, num_envs=n)
    ref = oracle.init_random(n, seed=31)
    for step in range(120):
        act = random_legal_actions(rng, ref["legal_action_mask"])
        act[rng.random(n) < 0.35] = 0
        st = env.step(st, torch.from_numpy(act), autoreset=True, inplace=(step % 2 == 0))
        oracle.step(ref, act, autoreset=True, seed=31)
        if step % 10 == 0 or step > 110:
            assert_state_equal(st, ref, where=f"K={k} autoreset step {step}")
    assert ref["board_ctr"].min() >= 1


def test_explicit_deals_wb5_auction(env, oracle, dds):
    # wb5/utils.py:61-75
    auction = [0, 9, 11, 20, 1, 0, 22, 1, 2, 0, 0, 28, 0, 0]
    n = 40
    hands = np.stack([oracle.key_to_hand(dds["keys"][i]) for i in range(n)])
    tricks = dds["tricks"][:n].reshape(n, 20)
    st = env.init_from_deals(hands, 1, False, False, [0, 3, 1, 2], tricks)
    ref = oracle.init_explicit(hands, 1, 0, 0, [0, 3, 1, 2], tricks)
    assert_state_equal(st, ref, where="explicit init")
    for i, a in enumerate(auction + [0, 0]):
        st = env.step(st, torch.full((n,), a, dtype=torch.int32))
        oracle.step(ref, np.full(n, a, np.int32))
        assert_state_equal(st, ref, where=f"wb5 call {i}")
    assert ref["terminated"].all()


def test_illegal_action_matches_oracle(env, oracle, dds):
    n = 64
    st = env.init(3, num_envs=n)
    ref = oracle.init_random(n, seed=3)
    for a in ([10] * n, [5] * n):  # second call bids BELOW the first: illegal
        st = env.step(st, torch.tensor(a, dtype=torch.int32))
        oracle.step(ref, np.array(a, np.int32))
    assert ref["illegal"].all()
    assert_state_equal(st, ref, fields={"terminated", "rewards", "illegal", "legal_action_mask", "current_player", "observation"},
                       where="illegal")


def test_observe_any_player(env, oracle):
    n = 300
    rng = np.random.default_rng(5)
    st = env.init(8, num_envs=n)
    ref = oracle.init_random(n, seed=8)
    for _ in range(9):
        act = random_legal_actions(rng, ref["legal_action_mask"])
        st = env.step(st, torch.from_numpy(act))
        oracle.step(ref, act)
    import brl_amd
    for p in range(4):
        got = to_np(brl_amd._observe(st, torch.full((n,), p, dtype=torch.int32)))
        assert np.array_equal(got, oracle.observe(ref, p))
    pid = rng.integers(0, 4, n).astype(np.int32)
    assert np.array_equal(to_np(env.observe(st, torch.from_numpy(pid))), oracle.observe(ref, pid))
    pos = to_np(brl_amd._player_position(torch.from_numpy(pid), st))
    assert np.array_equal(pos, np.array([list(ref["shuffled_players"][i]).index(pid[i]) for i in range(n)]))


@pytest.mark.parametrize("k,ws,substeps,n,T", [
    (1, "0", 1, 256, 32), (2, "0", 1, 515, 16), (4, "0", 1, 2048, 32), (8, "0", 1, 1000, 40),
    (4, "0", 4, 1024, 32), (8, "0", 4, 333, 12), (1, "0", 4, 64, 8),
    (4, None, 1, 128, 32),   # BASELINE.json configs[0]: num_envs=128, num_steps=32, random policy
    (4, None, 1, 2048, 32), (4, None, 4, 1000, 16), (4, None, 1, 1, 5), (4, None, 1, 33, 32),
    (4, None, 1, 4099, 33), (4, None, 4, 515, 12), (4, None, 1, 33, 7), (4, None, 1, 300, 64), (4, None, 1, 129, 7),
    (4, None, 4, 2048, 40), (4, None, 1, 2, 3), (4, None, 1, 30, 33), (4, None, 3, 700, 21), (4, None, 2, 450, 19),
    (4, None, 8, 130, 5), (4, None, 9, 130, 3),
    # the same shapes as the k_rollout_fs cases above (128x32, 2048x32), forced onto k_rollout_ws; more k_rollout_fs shapes
    (4, "ws", 1, 128, 32), (4, "ws", 1, 2048, 32), (4, None, 1, 32, 1), (4, None, 1, 64, 40), (4, None, 1, 96, 3),
    (4, None, 1, 4096, 33), (4, None, 1, 160, 9), (4, None, 1, 8192, 32),
    # the competitive macro-step with every seat random (src/utils.py:69-128) at the BASELINE size and at block-sized shapes
    (4, None, 4, 8192, 32), (4, None, 4, 32, 1), (4, None, 4, 96, 11), (4, None, 4, 64, 10), (4, None, 2, 64, 7),
    (4, None, 2, 4096, 21)])
def test_fused_random_rollout_matches_oracle(dds, oracle, k, ws, substeps, n, T):
    import brl_amd
    env = make_env(dds, k, ws)
    cfg = {"num_steps": T, "game_mode": "competitive" if substeps == 4 else "normal", "substeps": substeps,
           "reward_scale": 7600}
    roll = brl_amd.make_random_roll_out(cfg, env)
    st = env.init(2024, num_envs=n)
    ref = oracle.init_random(n, seed=2024)
    rs = (None, None, st, st.observation, 0, 0)
    draw = 0
    for call in range(2):  # two back-to-back rollouts: state and draw counter carry over
        rs, traj = roll(rs)
        want = oracle.rollout_random(ref, T, seed=2024, substeps=substeps, draw_base=draw)
        draw += T * substeps
        torch.cuda.synchronize()
        for name in ("obs", "legal_action_mask", "action", "done", "value", "reward", "log_prob"):
            g, o = to_np(getattr(traj, name)), want[name]
            assert g.shape == o.shape and np.array_equal(g, o), f"K={k} ws={ws} sub={substeps} call {call}: {name}"
        assert_state_equal(rs[2], ref, where=f"rollout final state K={k} sub={substeps} call {call}")
        assert np.array_equal(to_np(rs[3]), ref["observation"])
        assert rs[5] == draw


def test_rollout_terminated_count_accumulates(env, oracle):
    import brl_amd
    n, T = 512, 32
    roll = brl_amd.make_random_roll_out({"num_steps": T}, env)
    st = env.init(6, num_envs=n)
    ref = oracle.init_random(n, seed=6)
    rs = (None, None, st, None, 0, 0)
    rs, _ = roll(rs)
    a = oracle.rollout_random(ref, T, seed=6)["terminated_count"]
    rs, _ = roll(rs)
    b = oracle.rollout_random(ref, T, seed=6, draw_base=T)["terminated_count"]
    assert int(rs[4].item()) == a + b and a > 0


@pytest.mark.parametrize("n,T,gamma,lam", [(8192, 32, 1.0, 0.95), (64, 7, 0.99, 0.9), (32, 1, 1.0, 1.0), (2048, 40, 0.97, 0.8),
                                           (96, 64, 0.99, 0.95), (50, 12, 1.0, 0.9)])  # the last two: the two-launch shapes
def test_rollout_random_gae_in_one_launch(env, oracle, dds, n, T, gamma, lam):
    """brl_rollout_random_gae: the Transition of brl_rollout_random AND calc_gae's advantages / targets of that very
    trajectory from one launch — every column and the final state equal to the two-launch path, advantages / targets
    bit-identical to brl_gae and to the oracle's scan."""
    import ctypes as C
    import brl_amd
    from brl_amd import _capi
    from brl_amd.gae import gae_scan
    from brl_amd.roll_out import alloc_transition
    e = make_env(dds, 4)
    dev = e.device
    rng = np.random.default_rng(n + T)
    last_val = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(dev)
    gl = float(torch.tensor(gamma * lam, dtype=torch.float32))
    outs = []
    for fused in (False, True):
        st = e.init(77, num_envs=n)
        traj = alloc_transition(T, n, dev)
        p = _capi.TransitionPtrs()
        for f in _capi.TransitionPtrs._names:
            setattr(p, f, getattr(traj, f).data_ptr())
        lo = torch.empty((n, 480), dtype=torch.bool, device=dev); lm = torch.empty((n, 38), dtype=torch.bool, device=dev)
        tc = torch.zeros(1, dtype=torch.int64, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        if fused:
            adv = torch.empty((T, n), device=dev); tgt = torch.empty((T, n), device=dev)
            _capi.check(_capi.lib().brl_rollout_random_gae(e._h, st.packed.data_ptr(), n, T, 5, 7600.0, C.byref(p), lo.data_ptr(),
                                                           lm.data_ptr(), tc.data_ptr(), last_val.data_ptr(), gamma, gl,
                                                           adv.data_ptr(), tgt.data_ptr(), s))
        else:
            _capi.check(_capi.lib().brl_rollout_random(e._h, st.packed.data_ptr(), n, T, 1, 5, 7600.0, C.byref(p), lo.data_ptr(),
                                                       lm.data_ptr(), tc.data_ptr(), s))
            adv, tgt = gae_scan(e, traj.done, traj.value, traj.reward, last_val, gamma, lam)
        torch.cuda.synchronize()
        outs.append(list(traj) + [lo, lm, tc, st.packed, adv, tgt])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    want_adv, want_tgt = oracle.gae(to_np(outs[1][0]).astype(np.uint8), to_np(outs[1][2]), to_np(outs[1][3]), to_np(last_val), gamma, lam)
    assert np.array_equal(to_np(outs[1][-2]), want_adv) and np.array_equal(to_np(outs[1][-1]), want_tgt)
    assert n * T < 4096 or (bool(outs[1][0].any()) and float(outs[1][-2].abs().max()) > 0)


def test_host_mirror_of_the_one_launch_rollout_with_gae(dds):
    """brl_amd.make_random_roll_out_with_gae == make_random_roll_out followed by gae_scan (runner state included), twice in a row."""
    import brl_amd
    from brl_amd.gae import gae_scan
    cfg = {"num_steps": 16, "gamma": 0.99, "gae_lambda": 0.9}
    e1, e2 = make_env(dds, 4), make_env(dds, 4)
    r1, r2 = brl_amd.make_random_roll_out(cfg, e1), brl_amd.make_random_roll_out_with_gae(cfg, e2)
    rs1 = (None, None, e1.init(3, num_envs=128), None, 0, 9)
    rs2 = (None, None, e2.init(3, num_envs=128), None, 0, 9)
    lv = torch.linspace(-2.0, 2.0, 128, device=e1.device)
    for _ in range(2):
        rs1, t1 = r1(rs1)
        a1, g1 = gae_scan(e1, t1.done, t1.value, t1.reward, lv, 0.99, 0.9)
        rs2, t2, a2, g2 = r2(rs2, lv)
        for x, y in zip(list(t1) + [a1, g1, rs1[2].packed, rs1[3], rs1[4]], list(t2) + [a2, g2, rs2[2].packed, rs2[3], rs2[4]]):
            assert torch.equal(x, y)
        assert rs1[5] == rs2[5]


def test_gae_bit_exact(env, oracle):
    from brl_amd.gae import gae_scan
    rng = np.random.default_rng(2)
    for (T, N, gamma, lam) in ((32, 8192, 1.0, 0.95), (7, 130, 0.99, 0.9), (1, 1, 1.0, 1.0), (33, 1000, 0.97, 0.9),
                               (64, 17, 1.0, 0.95), (65, 40, 0.99, 0.95), (5, 16, 1.0, 0.95), (3, 64, 0.9, 0.8)):
        done = rng.random((T, N)) < 0.07
        value = rng.standard_normal((T, N)).astype(np.float32)
        reward = (rng.standard_normal((T, N)) * 0.2).astype(np.float32)
        last = rng.standard_normal(N).astype(np.float32)
        adv, tgt = gae_scan(env, torch.from_numpy(done).cuda(), torch.from_numpy(value).cuda(),
                            torch.from_numpy(reward).cuda(), torch.from_numpy(last).cuda(), gamma, lam)
        wa, wt = oracle.gae(done.astype(np.uint8), value, reward, last, gamma, lam)
        assert np.array_equal(to_np(adv), wa) and np.array_equal(to_np(tgt), wt)


def test_imp_reward_kats_and_oracle(env, oracle):
    import brl_amd
    from tests.test_oracle_kat import IMP_DOCTESTS
    for a, b, want in IMP_DOCTESTS:  # src/duplicate.py:20-43
        got = brl_amd._imp_reward(torch.tensor(a, dtype=torch.float32), torch.tensor(b, dtype=torch.float32), env=env)
        assert to_np(got).tolist() == [float(x) for x in want]
    rng = np.random.default_rng(4)
    a = (rng.integers(-760, 761, (5000, 1)) * 10 * np.array([1, 1, -1, -1])).astype(np.float32)
    b = (rng.integers(-760, 761, (5000, 1)) * 10 * np.array([1, 1, -1, -1])).astype(np.float32)
    got = to_np(brl_amd._imp_reward(torch.from_numpy(a), torch.from_numpy(b), env=env))
    want = np.stack([oracle.imp_reward(a[i], b[i]) for i in range(5000)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("k", [1, 4])
def test_duplicate_step_matches_oracle(dds, oracle, k):
    import brl_amd
    from oracle import Oracle
    env = make_env(dds, k)
    n = 901
    rng = np.random.default_rng(21)
    st = env.init(55, num_envs=n)
    ref = oracle.init_random(n, seed=55)
    A, B = brl_amd.Table_info.from_state(st), brl_amd.Table_info.from_state(st)
    oA, oB = Oracle.table_info_from(ref), Oracle.table_info_from(ref)
    step_fn = brl_amd.duplicate_step(env.step)
    cum = np.zeros(n, np.float32)
    it = 0
    while not ref["terminated"].all() and it < 700:
        act = random_legal_actions(rng, ref["legal_action_mask"])
        act[rng.random(n) < 0.4] = 0
        st, A, B = step_fn(st, torch.from_numpy(act), A, B)
        oracle.duplicate_step(ref, act, oA, oB)
        cum += ref["rewards"][:, 0]
        if it % 11 == 0:
            assert_state_equal(st, ref, where=f"duplicate it {it}")
        it += 1
    assert ref["terminated"].all()
    assert_state_equal(st, ref, where="duplicate end")
    for T, oT in ((A, oA), (B, oB)):  # G12: the snapshot is what "final contract" means
        for f in ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"):
            assert np.array_equal(to_np(getattr(T, f)).astype(np.float64), oT[f].astype(np.float64)), f
    assert (oA["terminated"] == 1).all() and (oB["terminated"] == 1).all()
    # G8: IMP emitted exactly once per board
    want = np.stack([oracle.imp_reward(oA["rewards"][i], oB["rewards"][i]) for i in range(n)])[:, 0]
    assert np.array_equal(cum, want)
    # G11: duplicate_init helper = seat swap on the same deal
    d = brl_amd.duplicate_init(env.init(55, num_envs=n))
    f0 = oracle.init_random(n, seed=55)
    assert np.array_equal(to_np(d._shuffled_players), f0["shuffled_players"][:, [1, 0, 3, 2]])
    assert np.array_equal(to_np(d._hand), f0["hand"]) and np.array_equal(to_np(d._dealer), f0["dealer"])


def _masked_log_softmax(logits, mask):
    l = np.where(mask.astype(bool), logits.astype(np.float64), -np.inf)
    m = l.max(1, keepdims=True)
    return l - m - np.log(np.exp(l - m).sum(1, keepdims=True))


def test_policy_step_argmax_and_sample(env, oracle):
    from brl_amd.utils import policy_step, MODE, SAMPLE
    n = 4096
    rng = np.random.default_rng(12)
    st = env.init(19, num_envs=n)
    ref = oracle.init_random(n, seed=19)
    for _ in range(5):
        act = random_legal_actions(rng, ref["legal_action_mask"])
        st = env.step(st, torch.from_numpy(act))
        oracle.step(ref, act)
    mask = ref["legal_action_mask"].copy()
    logits = rng.standard_normal((n, 38)).astype(np.float32) * 2
    lsm = _masked_log_softmax(logits, mask)
    lg = torch.from_numpy(logits).cuda()
    action = torch.empty(n, dtype=torch.int32, device="cuda")
    logp = torch.empty(n, dtype=torch.float32, device="cuda")
    # arg-max (pi.mode()) — exact
    out = torch.empty_like(st.packed)
    policy_step(env, st.packed, out, lg, MODE, 0, False, action=action, log_prob=logp)
    a = to_np(action)
    assert np.array_equal(a, np.where(mask.astype(bool), logits, -np.inf).argmax(1))
    assert np.allclose(to_np(logp), lsm[np.arange(n), a], atol=1e-5)  # fp32 log-softmax tolerance
    ref2 = ref.copy()
    oracle.step(ref2, a)
    from brl_amd.bridge_bidding import State
    assert_state_equal(State(env, out), ref2, where="policy_step(mode) next state")
    # sampling: always legal, log-prob of the sampled action, inverse-CDF consistency, frequencies
    counts = np.zeros((n, 38))
    for d in range(64):
        policy_step(env, st.packed, out, lg, SAMPLE, d, False, action=action, log_prob=logp)
        a = to_np(action)
        assert mask[np.arange(n), a].all()
        assert np.allclose(to_np(logp), lsm[np.arange(n), a], atol=1e-5)
        u = np.array([oracle.action_draw(19, e, d) >> 8 for e in range(0, n, 64)]) / 2.0 ** 24
        p = np.exp(lsm[::64])
        cdf = np.cumsum(p, 1)
        aa = a[::64]
        hi = cdf[np.arange(len(aa)), aa]
        lo = hi - p[np.arange(len(aa)), aa]
        assert ((u >= lo - 1e-5) & (u <= hi + 1e-5)).all()  # the draw falls in the chosen action's CDF cell
        counts[np.arange(n), a] += 1
    # pooled chi-square-ish check: empirical frequency of the most likely action tracks its probability
    top = lsm.argmax(1)
    emp = counts[np.arange(n), top].sum() / (64 * n)
    assert abs(emp - np.exp(lsm[np.arange(n), top]).mean()) < 0.01


def replay_policy_rollout(oracle, ref, traj, sub_actions, seed, reward_scale=7600.0, env_offset=0):
    """Replays the recorded actions of a policy-in-the-loop rollout (sub-step 1: traj.action, sub-steps 2-4:
    sub_actions) through the oracle's auto_reset(step) and returns the Transition columns the reference's _env_step
    would have stored (src/roll_out.py:63-103, src/utils.py:69-128): pre-step obs / mask of the acting player (G4),
    done = OR of the four terminated flags (G2), reward = (r1+r2+r3+r4)[actor] / reward_scale (G1), boards replaced
    mid-macro-step (G3).  `ref` ends as the post-rollout state (rewards / terminated of the last macro-step)."""
    act0, sub = to_np(traj.action), to_np(sub_actions)
    T, n = act0.shape
    want = {"obs": np.zeros((T, n, 480), np.uint8), "legal_action_mask": np.zeros((T, n, 38), np.uint8),
            "done": np.zeros((T, n), np.uint8), "reward": np.zeros((T, n), np.float32)}
    count = 0
    for t in range(T):
        want["obs"][t] = ref["observation"]
        want["legal_action_mask"][t] = ref["legal_action_mask"]
        actor = ref["current_player"].copy()
        racc = np.zeros((n, 4), np.float32)
        term = np.zeros(n, np.int32)
        for k in range(4):
            oracle.step(ref, act0[t] if k == 0 else sub[t, k - 1], autoreset=True, seed=seed, env_offset=env_offset)
            racc += ref["rewards"]
            term |= ref["terminated"]
        want["done"][t] = term
        want["reward"][t] = racc[np.arange(n), actor] / np.float32(reward_scale)
        count += int(term.sum())
    ref["rewards"] = racc          # src/utils.py:126-128
    ref["terminated"] = term
    want["terminated_count"] = count
    return want


POLICY_CFG = {"reward_scale": 7600, "game_mode": "competitive", "actor_illegal_action_mask": True,
              "gamma": 1.0, "gae_lambda": 0.95}


@pytest.mark.parametrize("n,T,graph,dt,calls", [(2048, 32, False, None, 2), (2048, 32, True, None, 2),
                                                (8192, 32, True, None, 1),     # configs[3]'s rollout at the reference's fp32
                                                (8192, 32, True, "x3", 2),     # ... its hidden layers on brl_mlp_gemm_x3 (inference_gemm = "bf16x3")
                                                (8192, 8, False, "x3", 1),     # (both: brl_linear_x3p — operands pre-split into planes, bf16 observations)
                                                (8192, 8, True, "x3s", 1),     # BRL_INFERENCE_PLANES=0: brl_mlp_gemm_x3, the split in registers
                                                (8192, 32, True, "bf16", 1), (1000, 9, False, "bf16", 2),
                                                (1000, 9, True, "fp16", 2)])
def test_policy_rollout_replays_through_oracle(env, oracle, n, T, graph, dt, calls, monkeypatch):
    """A7 with MLPs in the loop (BASELINE configs[2]/[3] rollout): every integer / byte column of the Transition and the
    final packed state bit-exact vs the oracle replay of the recorded actions; reward exact (integer scores / 7600 in
    fp32); value / log_prob vs an fp32 torch recomputation on the stored obs (tolerances stated below)."""
    import brl_amd
    from brl_amd.models import make_forward_pass
    gemm = "bf16x3" if dt in ("x3", "x3s") else None
    no_planes = dt == "x3s"
    if no_planes:
        monkeypatch.setenv("BRL_INFERENCE_PLANES", "0")
    planes = dt == "x3" or (dt is None and n >= 4096)      # (the default for fp32 forwards of >= 4096 rows)
    dt = None if dt in ("x3", "x3s") else dt
    cfg = dict(POLICY_CFG, num_steps=T, graph_rollout=graph, inference_dtype=dt, inference_gemm=gemm)
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(0, device="cuda"), fp.init(1, device="cuda")
    roll = brl_amd.make_roll_out(cfg, env, fp, fp)
    seed = 4242
    st = env.init(seed, num_envs=n)
    ref = oracle.init_random(n, seed=seed)
    rs = (actor, None, st, st.observation, 0, 0)
    total = 0
    for call in range(calls):
        rs, traj = roll(rs, opp)
        torch.cuda.synchronize()
        want = replay_policy_rollout(oracle, ref, traj, roll.sub_actions, seed)
        where = f"n={n} T={T} graph={graph} dt={dt} gemm={gemm} call {call}"
        for name in ("obs", "legal_action_mask", "done", "reward"):
            assert np.array_equal(to_np(getattr(traj, name)), want[name]), f"{where}: {name}"
        mask, act = want["legal_action_mask"], to_np(traj.action)
        assert np.take_along_axis(mask, act[..., None].astype(np.int64), 2).all(), f"{where}: sampled action illegal"
        assert_state_equal(rs[2], ref, where=f"{where}: final state")
        assert np.array_equal(to_np(rs[3]), ref["observation"])
        total += want["terminated_count"]
        assert int(rs[4].item()) == total and rs[5] == 4 * T * (call + 1)
        if dt is None:      # which layer kernel ran: the planes path takes its observations as bf16
            assert (roll.engine.xin.dtype == torch.bfloat16) == planes and (roll.engine.snap_actor.wp is not None) == (not no_planes)
        with torch.no_grad():
            logits, value = actor(traj.obs.reshape(T * n, 480).float())
        lsm = _masked_log_softmax(to_np(logits), mask.reshape(T * n, 38))
        # fp32 inference: accumulation order of the GEMMs differs (fused epilogue / merged heads): 2e-4; bf16: 8 mantissa bits
        # (bf16x3: fp32 operands as three exact bf16 pieces — fp32-grade: the same 2e-4)
        tol = 2e-4 if dt is None else (0.08 if dt == "bf16" else 0.01)   # (fp16: 11 mantissa bits)
        assert np.abs(to_np(traj.log_prob).reshape(-1) - lsm[np.arange(T * n), act.reshape(-1)]).max() < tol
        assert np.abs(to_np(traj.value).reshape(-1) - to_np(value)).max() < tol
    assert total > 0


@pytest.mark.parametrize("variant", ["free-run", "unmasked", "fair-tanh"])
def test_policy_rollout_variants_replay_through_oracle(env, oracle, variant):
    """free-run (G16: opponents pass, partner greedy), the unmasked / illegal-action-penalty policy
    (src/roll_out.py:33-39: the actor may draw an illegal call -> pgx penalty, board over) and a network outside
    InferenceSnapshot's coverage (FAIR, tanh) — all replayed through the oracle like the competitive rollout."""
    import brl_amd
    from brl_amd.models import make_forward_pass
    n, T, seed = 700, 12, 99
    cfg = dict(POLICY_CFG, num_steps=T)
    fp = make_forward_pass("relu", "DeepMind")
    if variant == "free-run":
        cfg["game_mode"] = "free-run"
    elif variant == "unmasked":
        cfg.update(actor_illegal_action_mask=False, actor_illegal_action_penalty=True)
    else:
        fp = make_forward_pass("tanh", "FAIR")
    actor, opp = fp.init(5, device="cuda"), fp.init(6, device="cuda")
    roll = brl_amd.make_roll_out(cfg, env, fp, fp)
    st = env.init(seed, num_envs=n)
    ref = oracle.init_random(n, seed=seed)
    rs, traj = roll((actor, None, st, st.observation, 0, 0), opp)
    torch.cuda.synchronize()
    want = replay_policy_rollout(oracle, ref, traj, roll.sub_actions, seed)
    for name in ("obs", "legal_action_mask", "done", "reward"):
        assert np.array_equal(to_np(getattr(traj, name)), want[name]), f"{variant}: {name}"
    assert_state_equal(rs[2], ref, where=f"{variant}: final state")
    act, sub, mask = to_np(traj.action), to_np(roll.sub_actions), want["legal_action_mask"]
    legal = np.take_along_axis(mask, act[..., None].astype(np.int64), 2)[..., 0]
    with torch.no_grad():
        logits, _ = actor(traj.obs.reshape(T * n, 480).float())
    if variant == "unmasked":
        assert (legal == 0).any()           # random weights: illegal calls do get drawn ...
        bad = legal == 0                    # ... and end the board at once with the offender's -1 (scaled)
        assert (to_np(traj.done)[bad] == 1).all()
        lsm = _masked_log_softmax(to_np(logits), np.ones((T * n, 38), np.uint8))   # log-prob under the UNMASKED softmax
    else:
        assert legal.all()
        lsm = _masked_log_softmax(to_np(logits), mask.reshape(T * n, 38))
    assert np.abs(to_np(traj.log_prob).reshape(-1) - lsm[np.arange(T * n), act.reshape(-1)]).max() < 2e-4
    if variant == "free-run":
        assert (sub[:, 0] == 0).all() and (sub[:, 2] == 0).all()   # both opponents always pass


def test_graphed_rollout_follows_reseed_and_lut_rotation(dds, oracle):
    """The captured launches read the RNG key / LUT through the library's device-resident context: after env.seed /
    env.set_lut (ppo.py:525-549) graph replays deal from the NEW table with the NEW key — checked against the oracle."""
    import brl_amd
    from brl_amd.models import make_forward_pass
    from oracle import Oracle
    n, T = 600, 6
    env = brl_amd.BridgeBidding(lut=(dds["keys"], dds["values"]))
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(0, device="cuda"), fp.init(1, device="cuda")
    roll = brl_amd.make_roll_out(dict(POLICY_CFG, num_steps=T, graph_rollout=True), env, fp, fp)
    st = env.init(5, num_envs=n)
    ref = oracle.init_random(n, seed=5)
    rs, traj = roll((actor, None, st, st.observation, 0, 0), opp)
    torch.cuda.synchronize()
    want = replay_policy_rollout(oracle, ref, traj, roll.sub_actions, 5)
    assert np.array_equal(to_np(traj.obs), want["obs"])
    k2, v2 = synthetic_lut(777, seed=9)
    env.set_lut((k2, v2))               # LUT rotation: new table, every env re-initialised (G14) with a new key
    orc2 = Oracle(k2, v2)
    st = env.init(6, num_envs=n)
    ref = orc2.init_random(n, seed=6)
    rs, traj = roll((actor, None, st, st.observation, rs[4], rs[5]), opp)
    torch.cuda.synchronize()
    want = replay_policy_rollout(orc2, ref, traj, roll.sub_actions, 6)
    for name in ("obs", "legal_action_mask", "done", "reward"):
        assert np.array_equal(to_np(getattr(traj, name)), want[name]), name
    assert_state_equal(rs[2], ref, where="graphed rollout after LUT rotation")
    assert int(rs[2]._lut_idx.max()) < 777 and int(rs[2]._board_count.max()) >= 1


def test_policy_rollout_with_mlp_is_self_consistent(env):
    import brl_amd
    from brl_amd.models import make_forward_pass
    n, T = 512, 8
    cfg = {"num_steps": T, "reward_scale": 7600, "game_mode": "competitive", "actor_illegal_action_mask": True,
           "gamma": 1.0, "gae_lambda": 0.95}
    fp = make_forward_pass("relu", "DeepMind")
    actor = fp.init(0, device="cuda")
    opp = fp.init(1, device="cuda")
    roll = brl_amd.make_roll_out(cfg, env, fp, fp)
    st = env.init(42, num_envs=n)
    rs = (actor, None, st, st.observation, 0, 0)
    rs2, traj = roll(rs, opp)
    torch.cuda.synchronize()
    obs, mask, act = to_np(traj.obs), to_np(traj.legal_action_mask), to_np(traj.action)
    assert obs.shape == (T, n, 480) and mask.shape == (T, n, 38)
    assert (obs[:, :, 428:].sum(-1) == 13).all()
    assert np.take_along_axis(mask, act[..., None].astype(np.int64), 2).all()  # sampled action always legal
    with torch.no_grad():
        logits, value = actor(traj.obs.reshape(T * n, 480).float())
    lsm = _masked_log_softmax(to_np(logits), mask.reshape(T * n, 38))
    assert np.allclose(to_np(traj.log_prob).reshape(-1), lsm[np.arange(T * n), act.reshape(-1)], atol=2e-4)
    assert np.allclose(to_np(traj.value).reshape(-1), to_np(value), atol=1e-5)
    r, d = to_np(traj.reward), to_np(traj.done)
    assert (r[d == 0] == 0).all() and np.abs(r).max() <= 4.0
    assert int(rs2[4].item()) == int(d.sum())
    assert np.array_equal(to_np(rs2[3]), to_np(rs2[2].observation))
    # determinism: same seed, same weights -> same bytes
    st_b = env.init(42, num_envs=n)
    _, traj_b = roll((actor, None, st_b, st_b.observation, 0, 0), opp)
    assert torch.equal(traj.action, traj_b.action) and torch.equal(traj.obs, traj_b.obs)
    adv, tgt = brl_amd.make_calc_gae(cfg, fp)(rs2, traj)
    assert adv.shape == (T, n) and torch.isfinite(adv).all() and torch.allclose(tgt, adv + traj.value)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_policy_rollout_low_precision_inference(env, dt):
    """Opt-in bf16 / fp16 inference (weights cast once per rollout, bias + ReLU in the GEMM epilogue, both heads in
    one GEMM): sampled actions legal, stored log_prob / value equal to the fp32 recomputation within the rounding
    of the inference dtype (tolerances below), update-to-update weight changes picked up."""
    import brl_amd
    from brl_amd.models import make_forward_pass
    n, T = 1024, 6
    cfg = {"num_steps": T, "reward_scale": 7600, "game_mode": "competitive", "actor_illegal_action_mask": True,
           "inference_dtype": dt}
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(0, device="cuda"), fp.init(1, device="cuda")
    roll = brl_amd.make_roll_out(cfg, env, fp, fp)
    st = env.init(7, num_envs=n)
    rs2, traj = roll((actor, None, st, st.observation, 0, 0), opp)
    torch.cuda.synchronize()
    mask, act = to_np(traj.legal_action_mask), to_np(traj.action)
    assert np.take_along_axis(mask, act[..., None].astype(np.int64), 2).all()
    with torch.no_grad():
        logits, value = actor(traj.obs.reshape(T * n, 480).float())
    lsm = _masked_log_softmax(to_np(logits), mask.reshape(T * n, 38))
    tol = 0.08 if dt == "bf16" else 0.02  # 8 / 11 mantissa bits through 5 layers of 1024-wide dot products
    assert np.abs(to_np(traj.log_prob).reshape(-1) - lsm[np.arange(T * n), act.reshape(-1)]).max() < tol
    assert np.abs(to_np(traj.value).reshape(-1) - to_np(value)).max() < tol
    # new weights -> new snapshot on the next call
    with torch.no_grad():
        for p in actor.parameters():
            p.mul_(0.5)
    _, traj2 = roll((actor, None, st, st.observation, 0, 0), opp)
    with torch.no_grad():
        _, value2 = actor(traj2.obs.reshape(T * n, 480).float())
    assert np.abs(to_np(traj2.value).reshape(-1) - to_np(value2)).max() < tol


def test_obs_cast_matches_torch(env):
    from brl_amd import _capi
    from brl_amd.bridge_bidding import _stream
    st = env.init(1, num_envs=777)
    obs = st.observation
    for fmt, dt in ((0, torch.float32), (1, torch.bfloat16), (2, torch.float16)):
        out = torch.empty(obs.shape, dtype=dt, device=obs.device)
        _capi.check(_capi.lib().brl_obs_cast(env._h, obs.data_ptr(), obs.shape[0], out.data_ptr(), fmt, _stream()))
        assert torch.equal(out, obs.to(dt))


@pytest.mark.parametrize("dt,fmt", [(torch.bfloat16, 1), (torch.float16, 2)])
def test_linear16_matches_float64_product(env, dt, fmt):
    """brl_linear_act (one hidden layer of the 16-bit inference path: hk.Linear + relu, src/models.py:23-33) against the float64
    product of the SAME 16-bit operands, to half an ulp of the 16-bit result + accumulation slack; shapes: the MLP's own
    (8192 x 1024 x 1024, K = 480), ragged M, K tails (480, 200, 8), a single chunk, M below one tile; strided operands; and
    against the library GEMM the default path would have used (same fp32-accumulate class: at most one 16-bit ulp apart)."""
    from brl_amd import _capi
    from brl_amd.bridge_bidding import _stream
    dev = env.device
    g = torch.Generator(device=dev).manual_seed(5)

    def run(x, w, b, y, relu):
        _capi.check(_capi.lib().brl_linear_act(env._h, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                                             b.data_ptr() if b is not None else None, y.data_ptr(), y.stride(0), x.shape[0],
                                             w.shape[0], x.shape[1], int(relu), fmt, _stream()))

    eps = 2.0 ** (-8 if fmt == 1 else -11)
    for (m, n, k) in ((8192, 1024, 1024), (8192, 1024, 480), (300, 256, 64), (257, 128, 8), (1000, 1024, 200), (5, 128, 1024),
                      (513, 384, 136)):
        x = (torch.rand(m, k, device=dev, generator=g) * 2 - 1).to(dt)
        if k == 480:
            x = (torch.rand(m, k, device=dev, generator=g) < 0.2).to(dt)   # an observation: 0 / 1
        w = (torch.randn(n, k, device=dev, generator=g) / k ** 0.5).to(dt)
        b = (torch.randn(n, device=dev, generator=g) * 0.1).to(dt).float()
        for relu in (1, 0):
            y = torch.full((m, n), float("nan"), device=dev).to(dt)
            run(x, w, b, y, relu)
            ref = x.double() @ w.double().t() + b.double()
            if relu:
                ref = ref.clamp_min(0)
            err = (y.double() - ref).abs()
            assert not torch.isnan(y.float()).any()
            assert int((err > ref.abs() * eps + 1e-3).sum()) == 0, (m, n, k, relu, float(err.max()))
        lib_y = torch.addmm(b.to(dt), x, w.t())
        y = torch.empty((m, n), dtype=dt, device=dev)
        run(x, w, b, y, 0)
        assert float((y.float() - lib_y.float()).abs().max()) <= float(lib_y.float().abs().max()) * 2 * eps
    # strided operands (row strides larger than the row), no bias
    xs = (torch.rand(700, 640, device=dev, generator=g) * 2 - 1).to(dt)
    ws = (torch.randn(256, 520, device=dev, generator=g) / 20).to(dt)
    ys = torch.zeros(700, 384, dtype=dt, device=dev)
    x, w, y = xs[:, :512], ws[:, :512], ys[:, :256]
    run(x, w, None, y, 0)
    ref = x.double() @ w.double().t()
    assert int(((y.double() - ref).abs() > ref.abs() * eps + 1e-3).sum()) == 0
    assert float(ys[:, 256:].abs().max()) == 0.0   # nothing written beyond column n_out
    # argument checks
    with pytest.raises(_capi.BrlError):
        run(xs[:, :512], ws[:200, :512], None, ys[:, :256], 0)     # n_out % 128
    with pytest.raises(_capi.BrlError):
        run(xs[:, :516], ws[:, :516], None, ys[:, :256], 0)        # k % 8


@pytest.mark.parametrize("dt,fmt", [(torch.bfloat16, 1), (torch.float16, 2)])
def test_heads_inside_the_launches_match_given_logits(env, dt, fmt):
    """The two ways a sub-step launch forms `actor(x), critic(x)` (src/models.py:30-33) itself, against the plain path:
      brl_macro_ext.head_part — partial products written by the last hidden layer's launch (brl_linear_act_heads, with and
        without the layer's own output being stored): parts + bias == the float64 product of the rounded layer output with the
        head weights (fp32 accumulation tolerance), the layer output equals brl_linear_act's bit for bit, and a sub-step launch
        on the parts == the same launch on logits summed on the host in the kernel's order (bit-identical action, log_prob,
        value, state, observation);
      brl_macro_ext.head_h — the heads formed from the stored layer output: same checks (logits to fp32 tolerance)."""
    import ctypes as C
    from brl_amd import _capi
    from brl_amd.bridge_bidding import _stream
    from brl_amd.utils import policy_step, SAMPLE
    L, dev = _capi.lib(), env.device
    g = torch.Generator(device=dev).manual_seed(21)
    n, hid = 1000, 1024
    st = env.init(23, num_envs=n)
    x = (torch.rand(n, hid, device=dev, generator=g) * 2 - 1).to(dt)
    w = (torch.randn(hid, hid, device=dev, generator=g) / hid ** 0.5 * 1.4).to(dt)
    b = (torch.randn(hid, device=dev, generator=g) * 0.1).to(dt).float()
    hw = (torch.randn(39, hid, device=dev, generator=g) / hid ** 0.5 * 3).to(dt)
    hb = (torch.randn(39, device=dev, generator=g) * 0.1).to(dt).float()
    y0 = torch.empty(n, hid, dtype=dt, device=dev)
    _capi.check(L.brl_linear_act(env._h, x.data_ptr(), hid, w.data_ptr(), hid, b.data_ptr(), y0.data_ptr(), hid, n, hid, hid, 1, fmt, _stream()))
    want = y0.double() @ hw.double().t() + hb.double()
    for store_y in (True, False):
        y = torch.zeros(n, hid, dtype=dt, device=dev)
        parts = torch.full((hid // 128, n, 40), float("nan"), device=dev)
        _capi.check(L.brl_linear_act_heads(env._h, x.data_ptr(), hid, w.data_ptr(), hid, b.data_ptr(), y.data_ptr() if store_y else None,
                                           hid, n, hid, hid, 1, fmt, hw.data_ptr(), hid, 39, parts.data_ptr(), 40, n * 40, _stream()))
        assert torch.equal(y.view(torch.int16), (y0 if store_y else torch.zeros_like(y0)).view(torch.int16))
        assert not torch.isnan(parts[:, :, :39]).any() and float(parts[:, :, 39].abs().max()) == 0.0   # (padding column: zeros)
        lg = hb.clone().expand(n, 39).contiguous()
        for p_ in range(parts.shape[0]):
            lg = lg + parts[p_, :, :39]          # the kernel's order
        assert float((lg.double() - want).abs().max()) < 2e-4 * float(want.abs().max())
    # ---- a sampled sub-step on the parts / on the stored layer output / on the logits themselves
    outs = {}
    for how in ("logits", "parts", "hidden"):
        o = {"state": torch.empty_like(st.packed), "action": torch.empty(n, dtype=torch.int32, device=dev),
             "logp": torch.empty(n, device=dev), "value": torch.empty(n, device=dev),
             "obs": torch.empty(n, 480, dtype=torch.bool, device=dev)}
        kw = dict(value_out=o["value"].data_ptr())
        if how == "logits":
            ext = _capi.MacroExt(value_in=lg[:, 38].data_ptr(), value_stride=lg.stride(0), **kw)
            policy_step(env, st.packed, o["state"], lg[:, :38], SAMPLE, 7, False, action=o["action"], log_prob=o["logp"], obs=o["obs"], ext=ext)
        elif how == "parts":
            ext = _capi.MacroExt(head_part=parts.data_ptr(), head_part_stride=parts.stride(0), head_part_ld=40, head_nparts=parts.shape[0],
                                 head_b=hb.data_ptr(), **kw)
            policy_step(env, st.packed, o["state"], None, SAMPLE, 7, False, action=o["action"], log_prob=o["logp"], obs=o["obs"], ext=ext)
        else:
            ext = _capi.MacroExt(head_h=y0.data_ptr(), head_ldh=hid, head_w=hw.data_ptr(), head_b=hb.data_ptr(), head_hidden=hid,
                                 head_fmt=fmt, **kw)
            policy_step(env, st.packed, o["state"], None, SAMPLE, 7, False, action=o["action"], log_prob=o["logp"], obs=o["obs"], ext=ext)
        outs[how] = o
    for k in outs["logits"]:
        assert torch.equal(outs["parts"][k], outs["logits"][k]), k
    # more than 8 parts (a wider last layer): the launch adds parts 8.. behind the first eight, in order
    p12 = torch.randn(12, n, 40, device=dev, generator=g) * 0.5
    lg12 = hb.clone().expand(n, 39).contiguous()
    for p_ in range(12):
        lg12 = lg12 + p12[p_, :, :39]
    a12, a_ref = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    lp12, lp_ref = torch.empty(n, device=dev), torch.empty(n, device=dev)
    scratch = torch.empty_like(st.packed)
    policy_step(env, st.packed, scratch, None, SAMPLE, 9, False, action=a12, log_prob=lp12,
                ext=_capi.MacroExt(head_part=p12.data_ptr(), head_part_stride=p12.stride(0), head_part_ld=40, head_nparts=12,
                                   head_b=hb.data_ptr()))
    policy_step(env, st.packed, scratch, lg12[:, :38], SAMPLE, 9, False, action=a_ref, log_prob=lp_ref, ext=_capi.MacroExt())
    assert torch.equal(a12, a_ref) and torch.equal(lp12, lp_ref)
    # the in-launch product sums in another order: same draws, logits equal to fp32 rounding -> almost every action agrees
    assert float((outs["hidden"]["value"] - outs["logits"]["value"]).abs().max()) < 2e-4 * float(want.abs().max())
    assert float((outs["hidden"]["action"] == outs["logits"]["action"]).float().mean()) > 0.995


def test_live_index_and_row_cast(env):
    """brl_live_index (the evaluators' loop condition as data) and brl_obs_cast_rows (gather + astype of the boards still playing)
    against torch: counts, ascending live indices, untouched tail; ragged n; every format."""
    from brl_amd import _capi
    from brl_amd.bridge_bidding import _stream
    L, dev = _capi.lib(), env.device
    g = torch.Generator(device=dev).manual_seed(3)
    for n in (1, 5, 1023, 1024, 1025, 8192, 10000):
        for p in (0.0, 0.3, 0.97, 1.0):
            term = torch.rand(n, device=dev, generator=g) < p
            live = torch.full((n,), -7, dtype=torch.int64, device=dev)
            fin = torch.zeros(1, dtype=torch.int64, device=dev)
            _capi.check(L.brl_live_index(env._h, term.data_ptr(), n, live.data_ptr(), fin.data_ptr(), -1, _stream()))
            want = (~term).nonzero().squeeze(1)
            assert int(fin) == int(term.sum())
            assert torch.equal(live[:want.numel()], want) and bool((live[want.numel():] == -7).all())
            _capi.check(L.brl_live_index(env._h, term.data_ptr(), n, None, fin.data_ptr(), -1, _stream()))   # count only
            assert int(fin) == int(term.sum())
    # the count straight into pinned host memory, tagged: the host polls the word (no event, no copy)
    import time
    word = torch.zeros(1, dtype=torch.int64).pin_memory()
    view = word.numpy()
    for tag, n in ((1, 777), (2, 8192), (0x7FFFFFFF, 10000)):
        term = torch.rand(n, device=dev, generator=g) < 0.4
        _capi.check(L.brl_live_index(env._h, term.data_ptr(), n, None, word.data_ptr(), tag, _stream()))
        t0 = time.perf_counter()
        while int(view[0]) >> 32 != tag:
            assert time.perf_counter() - t0 < 20.0, "the tagged count never arrived"
        assert int(view[0]) & 0xFFFFFFFF == int(term.sum())
    torch.cuda.synchronize()
    st = env.init(5, num_envs=3000)
    obs = st.observation
    rows = torch.randint(0, 3000, (777,), device=dev, generator=g)
    for fmt, dt in ((0, torch.float32), (1, torch.bfloat16), (2, torch.float16)):
        out = torch.empty((777, 480), dtype=dt, device=dev)
        _capi.check(L.brl_obs_cast_rows(env._h, obs.data_ptr(), rows.data_ptr(), 777, out.data_ptr(), fmt, _stream()))
        assert torch.equal(out, obs[rows].to(dt))


def test_eval_step_team_writes_the_float_observation_too(env):
    """brl_eval_step_team.obs_f32: the new observation as the next forward's float32 input, by the launch that steps the boards —
    equal to `observation.astype(float32)` (src/evaluation.py:52) for acting, waiting and finished boards alike."""
    from brl_amd import _capi
    from brl_amd.bridge_bidding import _stream
    L, dev, n = _capi.lib(), env.device, 3000
    st = env.init(9, num_envs=n)
    packed = st.packed.clone()
    g = torch.Generator(device=dev).manual_seed(1)
    for it in range(12):
        logits = torch.randn(n, 39, device=dev, generator=g)
        obs = torch.empty((n, 480), dtype=torch.bool, device=dev)
        x = torch.full((n, 480), -1.0, device=dev)
        term = torch.empty(n, dtype=torch.bool, device=dev)
        _capi.check(L.brl_eval_step_team(env._h, packed.data_ptr(), packed.data_ptr(), n, logits.data_ptr(), 39, it & 1, None, None, None, 0,
                                         None, None, None, obs.data_ptr(), None, None, term.data_ptr(), None, x.data_ptr(), _stream()))
        assert torch.equal(x, obs.to(torch.float32)), it
    assert bool(term.any())   # (some boards are finished by now: their rows were checked too)


def test_evaluators_on_live_rows_equal_full_batches(env, monkeypatch):
    """The evaluators forward only the boards still playing (brl_amd/evaluation.py::_ActiveRows): every returned number equals
    the run that forwards all boards every iteration."""
    from brl_amd.evaluation import make_simple_duplicate_evaluate, make_evaluate, make_evaluate_log
    from brl_amd.models import make_forward_pass
    fp = make_forward_pass("relu", "DeepMind")
    p1, p2 = fp.init(0, device="cuda"), fp.init(1, device="cuda")
    n = 3000
    res = {}
    for compact in ("1", "0"):
        monkeypatch.setenv("BRL_EVAL_COMPACT", compact)
        dup = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", n)
        full = make_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", p2, n, duplicate=True)
        def flat(x):
            if isinstance(x, (tuple, list)):
                return [v for y in x for v in flat(y)]
            return [float(v) for v in torch.as_tensor(x).double().reshape(-1)]
        res[compact] = (flat(dup(p1, p2, 3)), make_evaluate_log(full(p1, 4)[0]))
    # The calls are arg-maxes of logits whose per-row values could differ in the last bits with the batch size (the library may pick
    # another GEMM kernel for another M): a tie within that noise would change ONE board's auction.  So: equal up to what one
    # board of 3000 can move (in the runs so far every number was identical).
    a, b = res["1"], res["0"]
    assert len(a[0]) == len(b[0]) and all(abs(x - y) <= 0.02 + 1e-3 * abs(y) for x, y in zip(a[0], b[0])), (a[0], b[0])
    assert a[1].keys() == b[1].keys()
    assert all(abs(a[1][k] - b[1][k]) <= 0.02 + 1e-3 * abs(b[1][k]) for k in a[1]), {k: (a[1][k], b[1][k]) for k in a[1] if a[1][k] != b[1][k]}


@pytest.mark.parametrize("dt", [None, "bf16"])
def test_graphed_policy_rollout_matches_eager(env, dt):
    """config["graph_rollout"]: every macro-step replayed from a hipGraph (device-side draw index,
    brl_policy_step_at) — same bytes as the eager loop, call after call, also after a weight update."""
    import brl_amd
    from brl_amd.models import make_forward_pass
    n, T = 640, 5
    base = {"num_steps": T, "reward_scale": 7600, "game_mode": "competitive", "actor_illegal_action_mask": True,
            "inference_dtype": dt}
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(3, device="cuda"), fp.init(4, device="cuda")
    eager = brl_amd.make_roll_out(base, env, fp, fp)
    graphed = brl_amd.make_roll_out(dict(base, graph_rollout=True), env, fp, fp)
    st = env.init(11, num_envs=n)
    rs_e = rs_g = (actor, None, st, st.observation, 0, 0)
    for call in range(3):
        rs_e, tr_e = eager(rs_e, opp)
        rs_g, tr_g = graphed(rs_g, opp)
        torch.cuda.synchronize()
        for name in tr_e._fields:
            assert torch.equal(getattr(tr_e, name), getattr(tr_g, name)), f"{dt} call {call}: {name}"
        assert torch.equal(rs_e[2].packed, rs_g[2].packed) and torch.equal(rs_e[3], rs_g[3])
        assert int(rs_e[4].item()) == int(rs_g[4].item()) and rs_e[5] == rs_g[5]
        assert torch.equal(rs_e[2].rewards, rs_g[2].rewards) and torch.equal(rs_e[2].terminated, rs_g[2].terminated)
        with torch.no_grad():  # "update": the next call must see the new weights
            for p in actor.parameters():
                p.add_(0.01 * torch.randn_like(p))


def test_capture_with_a_dead_graph_cycle_and_an_eager_collector(env):
    """A dead reference cycle that still owns a hipGraph must be gone BEFORE the rollout's / the update's capture begins, and the
    cyclic collector must not run while a stream captures: the graph's destructor synchronises the device, which aborts the
    process mid-capture (brl_amd/_capture.py).  Seen as `Fatal Python error: Aborted ... Garbage-collecting` in
    test_ppo_iteration_at_config3_size[bf16], whose first parametrisation leaves such cycles behind."""
    import gc
    import weakref
    import brl_amd
    from brl_amd.models import make_forward_pass
    from brl_amd.train import DEFAULTS
    from brl_amd.update import FusedMinibatch, make_optimizer

    class Holder:
        pass

    def dead_cycle():
        h = Holder()
        h.me, h.graph, h.x = h, torch.cuda.CUDAGraph(), torch.zeros(64, device="cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            h.x.add_(1.0)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(h.graph):
            h.x.add_(1.0)
        h.graph.replay()
        return weakref.ref(h)

    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(3, device="cuda"), fp.init(4, device="cuda")
    n, T = 256, 4
    cfg = {"num_steps": T, "reward_scale": 7600, "game_mode": "competitive", "actor_illegal_action_mask": True, "graph_rollout": True}
    ucfg = dict(DEFAULTS, num_envs=n, num_steps=T, minibatch_size=256, update_epochs=1)
    st = env.init(5, num_envs=n)
    old = gc.get_threshold()
    gc.collect()
    gc.disable()
    try:
        # the collector is off: only the captures' own up-front collection can free the cycle
        ref = dead_cycle()
        assert ref() is not None
        roll = brl_amd.make_roll_out(cfg, env, fp, fp)
        rs, traj = roll((actor, None, st, st.observation, 0, 0), opp)
        assert ref() is None and not gc.isenabled()
        ref = dead_cycle()
        fm = FusedMinibatch(ucfg, actor, make_optimizer(ucfg, actor)["opt"], 256, torch.device("cuda"))
        assert ref() is None and not gc.isenabled() and fm.graph is not None
        # ... and with a collector that runs at every container allocation
        gc.set_threshold(1, 1, 1)
        gc.enable()
        roll2 = brl_amd.make_roll_out(cfg, env, fp, fp)
        rs2, traj2 = roll2((actor, None, st, st.observation, 0, 0), opp)
        fm2 = FusedMinibatch(ucfg, actor, make_optimizer(ucfg, actor)["opt"], 256, torch.device("cuda"))
        assert gc.isenabled() and fm2.graph is not None
    finally:
        gc.set_threshold(*old)
        gc.enable()
    torch.cuda.synchronize()
    for name in traj._fields:
        assert torch.equal(getattr(traj, name), getattr(traj2, name)), name
    assert int(traj.done.sum()) == int(rs[4].item()) == int(rs2[4].item())


def test_full_size_properties(dds):
    """BASELINE.json size (N=8192, T=32): size-independent properties of the fused rollout."""
    import brl_amd
    from bench import synthetic_lut as bench_lut
    keys, values = bench_lut(100_000, 0)   # the bench's own table: 100 000 rows (ppo.py:128 hash_size)
    env = brl_amd.BridgeBidding(lut=(keys, values))
    n, T = 8192, 32
    roll = brl_amd.make_random_roll_out({"num_steps": T}, env)
    st = env.init(0, num_envs=n)
    rs, traj = roll((None, None, st, None, 0, 0))
    torch.cuda.synchronize()
    obs = traj.obs
    assert bool((obs[:, :, 428:].sum(-1) == 13).all())                      # 13 own cards
    assert bool((obs[:, :, 0] ^ obs[:, :, 1]).all()) and bool((obs[:, :, 2] ^ obs[:, :, 3]).all())  # vul one-hot
    hist = obs[:, :, 8:428].reshape(T, n, 35, 3, 4)
    assert bool((hist.sum(-1) <= 1).all())                                   # one seat per (bid, event)
    assert bool((hist[:, :, :, 1].sum(-1) <= hist[:, :, :, 0].sum(-1)).all())  # doubled only if bid
    assert bool((hist[:, :, :, 2].sum(-1) <= hist[:, :, :, 1].sum(-1)).all())  # redoubled only if doubled
    m = traj.legal_action_mask
    assert bool(m[:, :, 0].all()) and not bool((m[:, :, 1] & m[:, :, 2]).any())
    assert bool(torch.gather(m, 2, traj.action.long()[..., None]).all())
    nl = m.sum(-1).float()
    assert torch.allclose(traj.log_prob, -torch.log(nl), atol=1e-6)
    assert bool((traj.reward[~traj.done] == 0).all()) and float(traj.reward.abs().max()) <= 1.0
    assert int(rs[4].item()) == int(traj.done.sum())
    # idempotence / determinism: same seed -> identical bytes
    st2 = env.init(0, num_envs=n)
    _, traj2 = roll((None, None, st2, None, 0, 0))
    for a, b in zip(traj, traj2):
        assert torch.equal(a, b)


def test_bench_configuration_against_the_oracle():
    """The benchmark's OWN step (bench.py: brl_rollout_random_gae at num_envs=8192, num_steps=32 on
    bench.synthetic_lut(100_000, 0), the ppo.py:128 table size) against oracle.rollout_random + oracle.gae: all seven
    Transition columns, last_obs / mask, every field of the packed state, terminated_count, advantages / targets —
    bit-exact, for two consecutive steps (state, draw counter and count carry over like the bench's)."""
    import ctypes as C
    import brl_amd
    from bench import LUT_LEN, NUM_ENVS, NUM_STEPS, synthetic_lut as bench_lut
    from brl_amd import _capi
    from brl_amd.bridge_bidding import State
    from brl_amd.roll_out import alloc_transition
    from oracle import Oracle
    keys, values = bench_lut(LUT_LEN, 0)
    e = brl_amd.BridgeBidding(lut=(keys, values))
    orc = Oracle(keys, values)
    n, T, dev = NUM_ENVS, NUM_STEPS, e.device
    st = e.init(0, num_envs=n)
    ref = orc.init_random(n, seed=0)
    traj = alloc_transition(T, n, dev)
    p = _capi.TransitionPtrs()
    for f in _capi.TransitionPtrs._names:
        setattr(p, f, getattr(traj, f).data_ptr())
    lo = torch.empty((n, 480), dtype=torch.bool, device=dev); lm = torch.empty((n, 38), dtype=torch.bool, device=dev)
    tc = torch.zeros(1, dtype=torch.int64, device=dev)
    adv = torch.empty((T, n), device=dev); tgt = torch.empty((T, n), device=dev)
    gl = float(torch.tensor(1.0 * 0.95, dtype=torch.float32))
    last_val = torch.zeros(n, dtype=torch.float32, device=dev)
    total = 0
    for step in range(2):
        _capi.check(_capi.lib().brl_rollout_random_gae(e._h, st.packed.data_ptr(), n, T, step * T, 7600.0, C.byref(p),
                                                       lo.data_ptr(), lm.data_ptr(), tc.data_ptr(), last_val.data_ptr(), 1.0, gl,
                                                       adv.data_ptr(), tgt.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        want = orc.rollout_random(ref, T, seed=0, draw_base=step * T)
        for name in brl_amd.Transition._fields:
            assert np.array_equal(to_np(getattr(traj, name)).astype(want[name].dtype), want[name]), f"step {step}: {name}"
        assert np.array_equal(to_np(lo), ref["observation"]) and np.array_equal(to_np(lm), ref["legal_action_mask"])
        assert_state_equal(State(e, st.packed), ref, where=f"bench step {step}: packed state")
        total += want["terminated_count"]
        assert int(tc.item()) == total
        wa, wt = orc.gae(want["done"], want["value"], want["reward"], np.zeros(n, np.float32), 1.0, 0.95)
        assert np.array_equal(to_np(adv), wa) and np.array_equal(to_np(tgt), wt), f"step {step}: advantages / targets"
    assert total > 2 * 8192 and float(adv.abs().max()) > 0


def test_flag_synchronised_rollout_equals_barrier_kernel_at_full_size():
    """k_rollout_fs (default at the BASELINE shape) and k_rollout_ws write the same bytes: 8192 x 32, 100 000-row LUT,
    three back-to-back rollouts (state, draw counter and terminated_count carry over)."""
    import brl_amd
    from bench import synthetic_lut as bench_lut
    lut = bench_lut(100_000, 0)
    outs = []
    for ws in (None, "ws"):
        env = make_env(None, 4, ws, lut=lut)
        roll = brl_amd.make_random_roll_out({"num_steps": 32}, env)
        rs = (None, None, env.init(5, num_envs=8192), None, 0, 0)
        got = []
        for _ in range(3):
            rs, traj = roll(rs)
            got.append([t.clone() for t in traj] + [rs[2].packed.clone(), rs[3].clone(), rs[4].clone()])
        outs.append(got)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert int(outs[0][-1][-1].item()) > 0


@pytest.mark.parametrize("T", [41, 64, 100])
def test_long_rollouts_run_as_pieces_of_the_flag_synchronised_kernel(dds, oracle, T):
    """T > 40 does not fit k_rollout_fs's command buffer: brl_rollout_random launches it in pieces that continue from each
    other's state / draw counter / terminated count.  Same bytes as k_rollout_ws in one launch, and as the oracle."""
    import brl_amd
    outs = []
    for ws in (None, "ws"):
        env = make_env(dds, 4, ws)
        roll = brl_amd.make_random_roll_out({"num_steps": T}, env)
        rs = (None, None, env.init(21, num_envs=96), None, 0, 5)
        rs, traj = roll(rs)
        rs, traj2 = roll(rs)
        outs.append([t.clone() for t in traj] + [t.clone() for t in traj2] + [rs[2].packed.clone(), rs[3].clone(), rs[4].clone()])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    ref = oracle.init_random(96, seed=21)
    want = oracle.rollout_random(ref, T, seed=21, draw_base=5)
    for i, name in enumerate(brl_amd.Transition._fields):
        g = to_np(outs[0][i])
        assert np.array_equal(g.astype(want[name].dtype), want[name]), name


def test_fused_rollout_kernels_agree_across_the_draw_counter_wrap(dds, oracle):
    """The action-draw index is a uint32 that wraps: k_rollout_fs precomputes the launch's draws per 4-draw Philox block,
    k_rollout_ws / k_rollout_random step through them — all three (and the oracle) must agree across 2^32."""
    import brl_amd
    outs = []
    for ws in (None, "ws", "0"):
        env = make_env(dds, 4, ws)
        roll = brl_amd.make_random_roll_out({"num_steps": 16}, env)
        rs, traj = roll((None, None, env.init(12, num_envs=64), None, 0, 2 ** 32 - 7))
        outs.append([t.clone() for t in traj] + [rs[2].packed.clone()])
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x, y)
    ref = oracle.init_random(64, seed=12)
    want = oracle.rollout_random(ref, 16, seed=12, draw_base=2 ** 32 - 7)
    assert np.array_equal(to_np(outs[0][1]), want["action"]) and np.array_equal(to_np(outs[0][5]), want["obs"])


@pytest.mark.parametrize("rng0", [2 ** 31 + 3, 2 ** 32 - 6])
def test_policy_rollout_draw_counter_wraps_mod_2_32(env, rng0):
    """ONE convention for the action-draw counter (include/brl_hip.h: uint32, wraps): the policy-in-the-loop rollout's
    device-resident counter holds `rng & 0xFFFFFFFF` — its first sampled actions equal a direct brl_policy_step call
    with that host draw index, also above 2^31 and across 2^32."""
    import brl_amd
    from brl_amd.models import make_forward_pass
    from brl_amd.utils import SAMPLE, policy_step
    n, T = 256, 3
    fp = make_forward_pass("relu", "DeepMind")
    actor = fp.init(0, device="cuda")
    roll = brl_amd.make_roll_out(dict(POLICY_CFG, num_steps=T, graph_rollout=False), env, fp, fp)
    st = env.init(77, num_envs=n)
    packed0 = st.packed.clone()
    with torch.no_grad():
        logits, _ = actor(st.observation.float())
    rs, traj = roll((actor, None, st, st.observation, 0, rng0), actor)
    action = torch.empty(n, dtype=torch.int32, device="cuda")
    policy_step(env, packed0, torch.empty_like(packed0), logits, SAMPLE, rng0 & 0xFFFFFFFF, True, action=action)
    torch.cuda.synchronize()
    # (the rollout's forward uses fused-epilogue GEMMs: a draw within ~1e-6 of a CDF boundary may fall the other way; a
    #  WRONG draw index agrees on well under half of the tables)
    assert float((action == traj.action[0]).float().mean()) >= 0.98
    assert rs[5] == rng0 + 4 * T


@pytest.mark.parametrize("n", [640, 8192])   # 8192 = BASELINE.json configs[2] (num_envs=8192 duplicate self-play)
def test_simple_duplicate_evaluate_replays_through_oracle(env, oracle, n):
    """Config 3 driver (src/evaluation.py:69-204): record the greedy actions the two MLPs chose on the
    GPU, replay them through the oracle's duplicate_step, compare IMPs / final contracts (G8, G12)."""
    import brl_amd
    from brl_amd.evaluation import make_simple_duplicate_evaluate
    from brl_amd.models import make_forward_pass
    from oracle import Oracle
    fp = make_forward_pass("relu", "DeepMind")
    t1, t2 = fp.init(3, device="cuda"), fp.init(4, device="cuda")
    rec = []
    ev = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", n, sync_every=8, record_actions=rec)
    (mean, se, win), A, B = ev(t1, t2, 99)
    torch.cuda.synchronize()
    ref = oracle.init_random(n, seed=99)
    oA, oB = Oracle.table_info_from(ref), Oracle.table_info_from(ref)
    cum = np.zeros(n, np.float32)
    for a in rec:
        act = to_np(a)
        live = ref["terminated"] == 0
        assert (ref["legal_action_mask"][np.arange(n), act][live] == 1).all()  # greedy action is legal
        oracle.duplicate_step(ref, act, oA, oB)
        cum += ref["rewards"][:, 0]
    assert ref["terminated"].all() and oA["terminated"].all() and oB["terminated"].all()
    for T, oT in ((A, oA), (B, oB)):
        for f in ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"):
            assert np.array_equal(to_np(getattr(T, f)).astype(np.float64), oT[f].astype(np.float64)), f
    assert abs(float(mean) - cum.mean()) < 1e-5                                  # fp32 reduction tolerance
    assert abs(float(se) - cum.std(ddof=1) / np.sqrt(n)) < 1e-5
    assert abs(float(win) - (cum > 0).mean()) < 1e-6
    # same network on both sides: every board is bid identically at both tables -> 0 IMP everywhere
    (mean0, _, win0), _, _ = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", 256)(t1, t1, 5)
    assert float(mean0) == 0.0 and float(win0) == 0.0


@pytest.mark.parametrize("n", [640, 8192])   # 8192 = BASELINE.json configs[2]
def test_simple_duplicate_evaluate_default_loop_replays_through_oracle(env, oracle, n):
    """The loop configs[2] actually runs — the teams alternate (a board whose player to act is on the other team waits: action -1),
    forwards on the boards still playing, the last boards through brl_mlp_forward_rows — replayed through the oracle's
    duplicate_step from its own recorded calls: waiting boards keep their state, every call of a live board is legal, the
    snapshots of both tables, the IMPs and the three returned numbers match (src/evaluation.py:120-202, G8, G12)."""
    from brl_amd.evaluation import make_simple_duplicate_evaluate
    from brl_amd.models import make_forward_pass
    from oracle import Oracle
    fp = make_forward_pass("relu", "DeepMind")
    t1, t2 = fp.init(3, device="cuda"), fp.init(4, device="cuda")
    calls = []
    ev = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", n, record_calls=calls)
    (mean, se, win), A, B = ev(t1, t2, 99)
    torch.cuda.synchronize()
    ref = oracle.init_random(n, seed=99)
    oA, oB = Oracle.table_info_from(ref), Oracle.table_info_from(ref)
    cum = np.zeros(n, np.float32)
    rows = np.arange(n)
    waited = 0
    for i, a in enumerate(calls):
        act = to_np(a)
        idle = act < 0
        live = (ref["terminated"] == 0) & ~idle
        team = (ref["current_player"] >> 1)
        assert (team[live] == (i & 1)).all() and (team[idle & (ref["terminated"] == 0)] != (i & 1)).all()   # who acted, who waited
        assert (ref["legal_action_mask"][rows, np.where(idle, 0, act)][live] == 1).all()
        keep = (ref[idle].copy(), oA[idle].copy(), oB[idle].copy())
        oracle.duplicate_step(ref, np.where(idle, 0, act).astype(np.int32), oA, oB)
        ref[idle], oA[idle], oB[idle] = keep
        cum[~idle] += ref["rewards"][~idle, 0]
        waited += int(idle.sum())
    assert ref["terminated"].all() and oA["terminated"].all() and oB["terminated"].all()
    assert 0 < waited <= 2 * n + n // 2     # a board waits at most one iteration at its start and one at the table switch
    for T, oT in ((A, oA), (B, oB)):
        for f in ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"):
            assert np.array_equal(to_np(getattr(T, f)).astype(np.float64), oT[f].astype(np.float64)), f
    assert abs(float(mean) - cum.mean()) < 1e-5
    assert abs(float(se) - cum.std(ddof=1) / np.sqrt(n)) < 1e-5
    assert abs(float(win) - (cum > 0).mean()) < 1e-6


def _replay_evaluate(oracle, rec_actions, rec_logits, seed, n, duplicate):
    """Replays make_evaluate's loop through the oracle with the recorded greedy actions / logits and restates its
    statistics with oracle/eval_stats.py (numpy, after src/evaluation.py:583-1031)."""
    from oracle import Oracle
    from oracle.eval_stats import StepLog
    ref = oracle.init_random(n, seed=seed)
    oA, oB = Oracle.table_info_from(ref), Oracle.table_info_from(ref)
    cum = np.zeros(n, np.float32)
    rsum = np.zeros((n, 4), np.float32)
    log = StepLog(n)
    for a_t, l_t in zip(rec_actions, rec_logits):
        act, lg = to_np(a_t), to_np(l_t)
        live = ref["terminated"] == 0
        masked = np.where(ref["legal_action_mask"].astype(bool), lg, -np.inf)
        assert np.array_equal(act[live], masked.argmax(1)[live])   # masked_pi.mode() of the acting team's network
        log.update(ref["terminated"], ref["current_player"], ref["legal_action_mask"], lg, act, bid_set=not duplicate)
        if duplicate:
            oracle.duplicate_step(ref, act, oA, oB)
        else:
            oracle.step(ref, act)
            rsum += ref["rewards"]
        cum += ref["rewards"][:, 0]
    assert ref["terminated"].all()
    return ref, oA, oB, cum, rsum, log


def _assert_log_info(got, want, names=None):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        g, w = to_np(g).astype(np.float64), np.asarray(w, np.float64)
        # counts / n in fp32 on both sides, sums of <= 8192 terms: 1e-6 absolute + 1e-5 relative
        assert g.shape == w.shape and np.allclose(g, w, rtol=1e-5, atol=1e-6), f"log_info[{i}]: {g} != {w}"


@pytest.mark.parametrize("n,game_mode", [(640, "competitive"), (2048, "competitive"), (300, "free-run")])
def test_duplicate_evaluate_with_statistics_matches_oracle(env, oracle, n, game_mode):
    """§8f-2 make_evaluate(duplicate=True) (src/evaluation.py:607-1032) + make_evaluate_log (:1035-1115): every one of
    the 23 log_info entries vs the numpy restatement fed by the oracle replay; Table_info snapshots bit-exact."""
    from brl_amd.evaluation import make_evaluate, make_evaluate_log
    from brl_amd.models import make_forward_pass
    from oracle.eval_stats import EVAL_LOG_KEYS, duplicate_log_info
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(3, device="cuda"), fp.init(4, device="cuda")
    ra, rl = [], []
    ev = make_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", opp, n, game_mode=game_mode, duplicate=True,
                       sync_every=8, record_actions=ra, record_logits=rl)
    log_info, A, B = ev(actor, 321)
    torch.cuda.synchronize()
    ref, oA, oB, cum, _, log = _replay_evaluate(oracle, ra, rl, 321, n, True)
    for T, oT in ((A, oA), (B, oB)):
        for f in ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"):
            assert np.array_equal(to_np(getattr(T, f)).astype(np.float64), oT[f].astype(np.float64)), f
    _assert_log_info(log_info, duplicate_log_info(cum, log, ref["step_count"], oA, oB))
    d = make_evaluate_log(log_info)
    assert list(d)[:19] == EVAL_LOG_KEYS and len(d) == 19 + 4 * 35
    assert d["eval/actor_bid_probs/1C"] == float(log_info[6][0]) and d["eval/opp_contract_probs/7NT"] == float(log_info[9][34])
    if game_mode == "free-run":
        assert d["eval/opp_pass_ratio"] == 1.0 and d["eval/opp_illegal_action_probs"] == 0.0


@pytest.mark.parametrize("n,game_mode,duplicate", [(640, "competitive", True), (2048, "competitive", True),
                                                   (300, "free-run", True), (640, "competitive", False)])
def test_team_alternating_evaluation_equals_lockstep_evaluation(env, n, game_mode, duplicate, monkeypatch):
    """With two different networks the evaluators run ONE forward per iteration and let the teams take turns
    (brl_eval_step_team); a recording run keeps the reference's lock-step loop (two forwards, every board steps), which
    the tests above replay through the oracle.  Both must produce the same Table_info snapshots and the same log_info.
    (Full batches in both runs — BRL_EVAL_COMPACT=0 — so that every GEMM has the same shape in both and the comparison can be
    exact; forwards on the live rows only are compared in test_evaluators_on_live_rows_equal_full_batches.)"""
    monkeypatch.setenv("BRL_EVAL_COMPACT", "0")
    from brl_amd.evaluation import make_evaluate, make_simple_duplicate_evaluate
    from brl_amd.models import make_forward_pass
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(3, device="cuda"), fp.init(4, device="cuda")
    outs = []
    for rec in (None, []):
        ev = make_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", opp, n, game_mode=game_mode, duplicate=duplicate,
                           sync_every=8, record_actions=rec)
        outs.append(ev(actor, 321))
    torch.cuda.synchronize()
    if duplicate:
        (la, A1, B1), (lb, A2, B2) = outs
        for T1, T2 in ((A1, A2), (B1, B2)):
            for f in ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"):
                assert torch.equal(getattr(T1, f), getattr(T2, f)), f
    else:
        (_, la), (_, lb) = outs
    assert len(la) == len(lb)
    for x, y in zip(la, lb):
        assert torch.equal(torch.as_tensor(x), torch.as_tensor(y))
    if duplicate and game_mode == "competitive":   # the simple duplicate evaluator (mean IMP, standard error, win rate)
        res = []
        for rec in (None, []):
            sd = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", n, record_actions=rec)
            res.append(sd(actor, opp, 99))
        for x, y in zip(res[0][0], res[1][0]):
            assert torch.equal(x, y)
        for f in ("rewards", "last_bid", "last_bidder"):
            assert torch.equal(getattr(res[0][2], f), getattr(res[1][2], f))


def _sharded_eval_rank(rank, world, port, out_dir, backend="gloo"):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU of the box
    out = _run_evaluators(shard=True)
    torch.save(out, os.path.join(out_dir, f"eval{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run_evaluators(shard):
    """the three evaluators of the ppo.py loop on 1001 boards (an uneven split over two ranks), two different networks"""
    import brl_amd
    from brl_amd.evaluation import make_evaluate, make_simple_duplicate_evaluate, make_simple_evaluate
    from brl_amd.models import make_forward_pass
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wb5_dds_1000.npz"))
    env = brl_amd.BridgeBidding(lut=(d["keys"], d["values"]))
    fp = make_forward_pass("relu", "DeepMind")
    a, b = fp.init(0, device="cuda"), fp.init(1, device="cuda")
    n = 1001
    (imp, se, win), _, _ = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", n, shard=shard)(a, b, 5)
    log_info, _, _ = make_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", b, n, duplicate=True, shard=shard)(a, 6)
    score = make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", b, n, shard=shard)(a, 7)
    flat = [float(imp), float(se), float(win), float(score)]
    for x in log_info:
        flat += [float(v) for v in torch.as_tensor(x).reshape(-1).cpu()]
    return flat


def test_sharded_evaluators_equal_the_single_process_evaluation(tmp_path):
    """ppo.py:366-381,461-484: under a process group the evaluators split their boards over the ranks (global board indices:
    the same deals) and all-reduce sums — every rank ends with the single-process statistics (mean IMP, SE, win rate, the 23
    log_info entries, the single-table score).  Two gloo ranks on this box's GPU; 1001 boards = 501 + 500."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_sharded_eval_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "eval0.pt"), torch.load(tmp_path / "eval1.pt")
    want = _run_evaluators(shard=None)
    assert r0 == r1                                            # all-reduced: identical on both ranks
    assert len(want) == len(r0) and np.allclose(r0, want, rtol=2e-5, atol=1e-6), np.abs(np.array(r0) - np.array(want)).max()
    assert r0[0] == pytest.approx(want[0], abs=1e-6) and want[1] > 0


def test_single_table_evaluate_with_statistics_matches_oracle(env, oracle):
    """make_evaluate(duplicate=False) (src/evaluation.py:229-605): 19-entry log_info, rewards accumulated on the state."""
    from brl_amd.evaluation import make_evaluate
    from brl_amd.models import make_forward_pass
    from oracle.eval_stats import single_log_info
    n = 900
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(8, device="cuda"), fp.init(9, device="cuda")
    ra, rl = [], []
    ev = make_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", opp, n, duplicate=False, sync_every=4,
                       record_actions=ra, record_logits=rl)
    state, log_info = ev(actor, 55)
    torch.cuda.synchronize()
    ref, _, _, cum, rsum, log = _replay_evaluate(oracle, ra, rl, 55, n, False)
    _assert_log_info(log_info, single_log_info(cum, log, ref))
    ref["rewards"] = rsum   # state.replace(rewards=rewards) (:582)
    assert_state_equal(state, ref, where="single-table evaluate final state")


def test_simple_evaluate_runs_and_is_deterministic(env):
    from brl_amd.evaluation import make_simple_evaluate
    from brl_amd.models import make_forward_pass
    fp = make_forward_pass("relu", "DeepMind")
    a, o = fp.init(1, device="cuda"), fp.init(2, device="cuda")
    ev = make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", o, 512)
    r1, r2 = float(ev(a, 11)), float(ev(a, 11))
    assert r1 == r2 and abs(r1) <= 7600.0


@pytest.mark.parametrize("n", [512, 10000])   # 10000 = num_eval_envs of ppo.py's per-iteration evaluation (ppo.py:366)
def test_simple_evaluate_replays_through_oracle(env, oracle, n):
    """src/evaluation.py:11-66 (called every iteration, ppo.py:366): the greedy actor against a fixed greedy opponent on single
    tables without auto-reset.  The four calls of every macro-step are recorded on the GPU and replayed through the oracle's
    step: R = sum over iterations of (r1 + r2 + r3 + r4)[player to act before the macro-step] (src/evaluation.py:53-60,
    src/utils.py:196-198) — scores are integers, so the per-board returns must be EXACT and the mean equal up to the fp32
    reduction; finished boards keep receiving calls that must be no-ops (G9)."""
    from brl_amd.evaluation import make_simple_evaluate
    from brl_amd.models import make_forward_pass
    fp = make_forward_pass("relu", "DeepMind")
    a, o = fp.init(1, device="cuda"), fp.init(2, device="cuda")
    rec = []
    ev = make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", o, n, record_actions=rec)
    got = float(ev(a, 11))
    torch.cuda.synchronize()
    assert got == float(make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", o, n)(a, 11))   # recording changes nothing
    ref = oracle.init_random(n, seed=11)
    R = np.zeros(n, np.float64)
    rows = np.arange(n)
    for calls in rec:
        calls = to_np(calls)
        actor = ref["current_player"].copy()
        racc = np.zeros((n, 4), np.float64)
        for k in range(4):
            live = ref["terminated"] == 0
            assert (ref["legal_action_mask"][rows, calls[k]][live] == 1).all(), "a greedy call is illegal"
            oracle.step(ref, calls[k])
            racc += ref["rewards"]
        R += racc[rows, actor]
    assert ref["terminated"].all() and len(rec) >= 2
    assert np.abs(R).max() <= 7600 and (R != 0).any() and np.array_equal(R, np.round(R))
    assert abs(got - R.mean()) <= 1e-6 * max(1.0, np.abs(R).mean()) * 8    # fp32 sum of n integers / n
    # the play continued until every board was finished and not much longer (the loop runs <= 2 iterations past the end)
    done_at = None
    ref2 = oracle.init_random(n, seed=11)
    for i, calls in enumerate(rec):
        for k in range(4):
            oracle.step(ref2, to_np(calls)[k])
        if done_at is None and ref2["terminated"].all():
            done_at = i
    assert done_at is not None and len(rec) - 1 - done_at <= 2


@pytest.mark.parametrize("n", [512, 10000])
def test_simple_evaluate_by_turn_replays_through_oracle(env, oracle, n):
    """The default loop of make_simple_evaluate — one call per iteration, the actor's network on even calls, the opponent's on odd
    ones — replayed through the oracle's step from the recorded calls: every call legal while the board is live, finished boards
    untouched (G9), a board's return = what the player who opened collected (src/evaluation.py:53-60 summed over the macro-steps)."""
    from brl_amd.evaluation import make_simple_evaluate
    from brl_amd.models import make_forward_pass
    fp = make_forward_pass("relu", "DeepMind")
    a, o = fp.init(1, device="cuda"), fp.init(2, device="cuda")
    calls = []
    got = float(make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", o, n, record_calls=calls)(a, 11))
    torch.cuda.synchronize()
    assert got == float(make_simple_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", o, n)(a, 11))   # live-row forwards: same number
    ref = oracle.init_random(n, seed=11)
    opener = ref["current_player"].copy()
    rows = np.arange(n)
    R = np.zeros(n, np.float64)
    done_at = None
    for i, c in enumerate(calls):
        c = to_np(c)
        live = ref["terminated"] == 0
        assert (ref["legal_action_mask"][rows, c][live] == 1).all(), f"call {i}: a greedy call is illegal"
        oracle.step(ref, c)
        R += ref["rewards"][rows, opener]
        if done_at is None and ref["terminated"].all():
            done_at = i
    assert done_at is not None and len(calls) - 1 - done_at <= 2      # the loop stops at most two calls after the last board
    assert np.abs(R).max() <= 7600 and (R != 0).any() and np.array_equal(R, np.round(R))
    assert abs(got - R.mean()) <= 1e-6 * max(1.0, np.abs(R).mean()) * 8


@pytest.mark.parametrize("actor_kind,opp_kind", [(("tanh", "DeepMind"), ("relu", "FAIR")), (("relu", "FAIR"), ("relu", "DeepMind_6"))])
def test_simple_evaluate_by_turn_equals_the_macro_step_loop(env, monkeypatch, actor_kind, opp_kind):
    """make_simple_evaluate one call per iteration on the boards still playing (the default) against the mirror of the reference's
    macro-step loop (src/evaluation.py:36-63), for the networks the fused snapshots do not cover (tanh: brl_mlp_forward_rows + the
    module; FAIR: the module itself) — the same return, exactly (scores are integers; the calls are arg-maxes)."""
    from brl_amd.evaluation import make_simple_evaluate
    from brl_amd.models import make_forward_pass
    a = make_forward_pass(*actor_kind).init(1, device="cuda")
    o = make_forward_pass(*opp_kind).init(2, device="cuda")
    ev = make_simple_evaluate(env, actor_kind[0], actor_kind[1], opp_kind[0], opp_kind[1], o, 1536)
    fast = float(ev(a, 21))
    monkeypatch.setenv("BRL_SIMPLE_EVAL_BY_TURN", "0")
    slow = float(ev(a, 21))
    assert fast == slow and abs(fast) <= 7600.0


def test_ppo_loop_runs_end_to_end(env, tmp_path):
    """BASELINE config 4 at toy size — the ppo.py:348-570 loop: evaluations, FSP pool behind the threshold gate,
    roll_out -> calc_gae -> update_step, the reference's log keys, LUT rotation after hash_size boards (G14),
    checkpoints + final params / opt_state."""
    from brl_amd import checkpoint as ckpt
    from brl_amd.train import DEFAULTS, train
    cfg = dict(DEFAULTS, num_envs=256, num_steps=8, total_timesteps=256 * 8 * 4, minibatch_size=512, update_epochs=2,
               lut_len=2000, synthetic_lut_files=2, hash_size=300, num_eval_envs=128, num_prioritized_envs=64,
               num_eval_step=2, lr=1e-4, ratio_model_zoo=1.0, prioritized_fictitious=True, log_path=str(tmp_path),
               exp_name="t", graph_rollout=True)
    logs = []
    rs, hist = train(cfg, log=logs.append)
    assert len(hist) == 4 and all(np.isfinite(h["train/total_loss"]) for h in hist)
    for key in ("train/score", "train/value_loss", "train/loss_actor", "train/illegal_action_loss", "train/policy_entropy",
                "train/clipflacs", "train/approx_kl", "train/lr", "train/imp_opp_before", "train/imp_opp_after",
                "board_num", "steps"):                                   # ppo.py:501-519
        assert key in hist[-1], key
    assert hist[-1]["steps"] == 256 * 8 * 4 and hist[0]["train/illegal_action_loss"] > 0
    assert "eval/IMP_reward" in hist[0] and "eval/IMP_reward" in hist[2] and "eval/IMP_reward" not in hist[1]
    assert abs(hist[-1]["train/imp_opp_after"]) <= 24
    assert any("hash_table_next" in h for h in hist)                    # >= 300 boards finish per update: the table rotates
    assert hist[0]["opponent"] == "latest" and any(h["opponent"].startswith("params-") for h in hist[1:])   # FSP pool in use
    pool = os.path.join(str(tmp_path), "t", "rl_params")
    assert ckpt.list_checkpoints(pool) == [f"params-{i:08}.pt" for i in (1, 2, 3, 4)]
    assert os.path.exists(os.path.join(pool, "opt_state-00000004.pt"))
    back = ckpt.load_params(os.path.join(pool, "params-00000004.pt"), "relu", "DeepMind", "cuda")
    for a, b in zip(back.parameters(), rs[0].parameters()):
        assert torch.equal(a, b)


def test_latest_opponent_is_a_snapshot_not_the_live_learner(tmp_path):
    """ppo.py:455-460: the "latest" opponent keeps the PRE-update weights (an immutable pytree in the reference), so
    imp_opp_after compares new against old.  With the opponent aliased to the live module it is identically 0 and the
    threshold_model_zoo gate can never trigger."""
    from brl_amd.train import DEFAULTS, train
    cfg = dict(DEFAULTS, num_envs=256, num_steps=8, total_timesteps=256 * 8 * 3, minibatch_size=512, update_epochs=4,
               lut_len=2000, synthetic_lut_files=1, num_eval_envs=64, num_prioritized_envs=1024, num_eval_step=100,
               lr=3e-3, ratio_model_zoo=0.0, log_path=str(tmp_path), exp_name="s", save_model=False)
    rs, hist = train(cfg, log=lambda s: None)
    assert all(h["opponent"] == "latest" for h in hist)
    assert all(h["opp_weight_delta"] > 0 for h in hist)               # the update moved the learner away from its opponent
    assert any(h["train/imp_opp_after"] != 0.0 for h in hist)          # new vs old on 1024 boards: not a self-match
    assert all(h["train/imp_opp_before"] == 0.0 for h in hist)         # a fresh snapshot against itself: every board ties


def _two_gpus_or_skip():
    """RCCL needs one GPU per rank.  torch.cuda.device_count() does not initialise the GPU, so the spawned ranks are still
    the first to touch their devices."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"RCCL path needs >= 2 GPUs (one per rank); this box has {n} — covered with gloo ranks sharing the device")


def _train_rank(rank, world, port, out_dir, backend="gloo"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), BRL_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from brl_amd.train import DEFAULTS, train
    cfg = dict(DEFAULTS, num_envs=256, num_steps=8, total_timesteps=world * 256 * 8 * 3, minibatch_size=512, update_epochs=2,
               lut_len=2000, synthetic_lut_files=2, hash_size=100_000, num_eval_envs=128, num_prioritized_envs=64,
               num_eval_step=2, lr=1e-4, ratio_model_zoo=1.0, log_path=out_dir, exp_name="t",   # ONE pool directory: rank 0 saves
               graph_rollout=True, grad_allreduce="sharded" if world == 4 else "flat")   # (both forms run the loop)
    def keep_first_rollout(i, runner_state, traj, roll_out):
        """iteration 0's Transition + sub-step actions + final packed state of THIS rank's shard, for the parent's oracle replay"""
        if i == 0:
            torch.cuda.synchronize()
            torch.save({"traj": tuple(x.cpu() for x in traj), "sub_actions": roll_out.sub_actions.cpu(),
                        "packed": runner_state[2].packed.cpu(), "terminated_count": int(runner_state[4].item()),
                        "env_offset": runner_state[2].env.env_offset}, os.path.join(out_dir, f"rollout{rank}.pt"))

    rs, hist = train(cfg, log=lambda *_: None, on_rollout=keep_first_rollout)
    flat = torch.cat([p.detach().reshape(-1) for p in rs[0].parameters()]).cpu()
    torch.save((flat, [h["train/total_loss"] for h in hist], type(rs[1].get("graphed")).__name__,
                getattr(rs[1].get("graphed"), "world", None), torch.cuda.current_device()), os.path.join(out_dir, f"rank{rank}.pt"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _replay_rank_rollout(tmp_path, rank, num_envs=256):
    """a rank's first rollout of `_train_rank` (its own shard: env_offset = rank * num_envs) replayed through the oracle: the
    Transition's obs / mask / done / reward, the final packed state and the board count, bit for bit"""
    import brl_amd
    from bench import synthetic_lut as bench_lut
    from brl_amd.bridge_bidding import State
    from oracle import Oracle
    d = torch.load(tmp_path / f"rollout{rank}.pt")
    assert d["env_offset"] == rank * num_envs
    keys, values = bench_lut(2000, 0)                      # train(): luts[rotation.current], file 0 of the synthetic tables
    orc = Oracle(keys, values)
    ref = orc.init_random(num_envs, seed=0, env_offset=rank * num_envs)   # DEFAULTS["seed"] = 0
    traj = brl_amd.Transition(*d["traj"])
    want = replay_policy_rollout(orc, ref, traj, d["sub_actions"], 0, env_offset=rank * num_envs)
    for name in ("obs", "legal_action_mask", "done", "reward"):
        assert np.array_equal(to_np(getattr(traj, name)), want[name]), f"rank {rank}: {name}"
    act = to_np(traj.action)
    assert np.take_along_axis(want["legal_action_mask"], act[..., None].astype(np.int64), 2).all()
    assert d["terminated_count"] == want["terminated_count"] > 0
    env = brl_amd.BridgeBidding(lut=(keys, values))
    assert_state_equal(State(env, d["packed"].to(env.device)), ref, where=f"rank {rank}: packed state after its first rollout")
    return to_np(traj.obs)


def test_ppo_loop_two_ranks(tmp_path):
    """BASELINE config 5 at toy size: the ppo.py loop on two ranks (own env shard each via env_offset, ONE all-reduce of the flat
    gradient — the default "flat" form —, the pool's opponent index broadcast from rank 0; gloo on
    this one-GPU box, RCCL on a node): both ranks end with IDENTICAL parameters although their shards differ — and each rank's
    first rollout IS its shard: replayed through the oracle at env_offset = rank * num_envs."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_train_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0[2] == "FusedMinibatch" and r0[3] == 2
    assert len(r0[1]) == 3 and all(np.isfinite(x) for x in r0[1] + r1[1])
    assert r0[1] != r1[1]                      # different shards: different losses ...
    assert torch.equal(r0[0], r1[0])           # ... same parameters (all-reduced gradients)
    obs0, obs1 = _replay_rank_rollout(tmp_path, 0), _replay_rank_rollout(tmp_path, 1)
    assert not np.array_equal(obs0, obs1)


def test_ppo_loop_four_ranks(tmp_path):
    """The same loop on FOUR gloo ranks sharing the GPU (a box of this pool allows at most 6 processes on its card: the pytest
    process + 4): the sharded gradient step at world 4 (4 buckets x 4 slices), the pool's opponent index broadcast to three
    followers, four env shards — identical parameters on every rank, and rank 3's first rollout replayed through the oracle at
    env_offset = 3 * 256."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_train_rank, args=(4, port, str(tmp_path)), nprocs=4, join=True, start_method="spawn")
    rs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(4)]
    assert all(r[2] == "FusedMinibatch" and r[3] == 4 for r in rs)
    for r in rs[1:]:
        assert torch.equal(rs[0][0], r[0]) and r[1] != rs[0][1]
    _replay_rank_rollout(tmp_path, 3)


def test_ppo_loop_two_ranks_rccl(tmp_path):
    """The same loop over RCCL ("nccl"), one GPU per rank — BASELINE configs[4]'s collective path (the flat gradient's
    all-reduce between FusedMinibatch's graphs, opponent-index broadcast, summed counters) on real xGMI links.
    SKIPPED with a reason on a box with fewer than 2 GPUs."""
    _two_gpus_or_skip()
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_train_rank, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0[2] == "FusedMinibatch" and r0[3] == 2 and (r0[4], r1[4]) == (0, 1)   # one device per rank
    assert len(r0[1]) == 3 and all(np.isfinite(x) for x in r0[1] + r1[1]) and r0[1] != r1[1]
    assert torch.equal(r0[0], r1[0])           # post-run cross-rank parameter equality (a mis-captured collective would break it)
    _replay_rank_rollout(tmp_path, 1)


@pytest.mark.parametrize("infer", [None, "bf16"])   # None = the reference's fp32 forwards (the default), bf16 = the opt-in path
def test_ppo_iteration_at_config3_size(tmp_path, infer):
    """BASELINE.json configs[3] at ITS size: one ppo.py iteration with num_envs=8192, num_steps=32, minibatch 1024,
    10 epochs (2560 minibatch steps), DeepMind MLP, hipGraph rollout + update — finite statistics, every env-step
    counted, the weights moved, the pool file written."""
    from brl_amd import checkpoint as ckpt
    from brl_amd.train import DEFAULTS, train
    cfg = dict(DEFAULTS, num_envs=8192, num_steps=32, total_timesteps=8192 * 32, minibatch_size=1024, update_epochs=10,
               lr=1e-5, evaluate=False, log_path=str(tmp_path), exp_name="c3", graph_rollout=True, inference_dtype=infer,
               hash_size=50_000)
    rs, hist = train(cfg, log=lambda s: None)
    h = hist[0]
    assert len(hist) == 1 and h["steps"] == 8192 * 32 and h["board_num"] > 8192       # ~100 k boards per rollout
    for key in ("train/total_loss", "train/value_loss", "train/loss_actor", "train/policy_entropy", "train/approx_kl",
                "train/clipflacs", "train/illegal_action_loss"):
        assert np.isfinite(h[key]), key
    assert 0.0 < h["train/policy_entropy"] < np.log(38) and 0.0 <= h["train/clipflacs"] <= 1.0
    assert "hash_table_next" in h                                                      # ~96 k boards > hash_size: table rotated (G14)
    fresh = __import__("brl_amd.models", fromlist=["make_forward_pass"]).make_forward_pass("relu", "DeepMind").init(0, device="cuda")
    moved = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(rs[0].parameters(), fresh.parameters()))
    assert 0.0 < moved <= 2560 * 1e-5 * 1.01                                            # 2560 Adam steps of at most lr = 1e-5 each
    assert ckpt.list_checkpoints(os.path.join(str(tmp_path), "c3", "rl_params")) == ["params-00000001.pt"]


def test_macro_step_matches_manual_composition(env, oracle):
    """A6 / G3: single_play_step_two_policy_commpetitive_deterministic == step + 3 x (forward, arg-max, step)
    with auto-reset, rewards summed and terminated OR-ed (src/utils.py:133-202); replayed through the oracle."""
    from brl_amd.evaluation import masked_mode
    from brl_amd.models import make_forward_pass
    from brl_amd.utils import auto_reset, single_play_step_two_policy_commpetitive_deterministic
    n = 1024
    fp = make_forward_pass("relu", "FAIR")
    actor, opp = fp.init(7, device="cuda"), fp.init(8, device="cuda")
    step_fn = single_play_step_two_policy_commpetitive_deterministic(auto_reset(env.step, env.init), fp, actor, fp, opp)
    st = env.init(77, num_envs=n)
    ref = oracle.init_random(n, seed=77)
    rng = np.random.default_rng(0)
    for it in range(12):
        a1 = random_legal_actions(rng, ref["legal_action_mask"])
        a1[rng.random(n) < 0.5] = 0
        # manual composition on a copy of the GPU state (env.step is itself checked against the oracle)
        man = env.step(st, torch.from_numpy(a1), autoreset=True)
        oracle.step(ref, a1, autoreset=True, seed=77)
        rsum = ref["rewards"].copy()
        term = ref["terminated"].copy()
        for k, net in enumerate((opp, actor, opp)):
            with torch.no_grad():
                logits, _ = fp.apply(net, man.observation.float())
            a = masked_mode(logits, man.legal_action_mask)
            man = env.step(man, a, autoreset=True)
            oracle.step(ref, to_np(a), autoreset=True, seed=77)
            rsum += ref["rewards"]
            term |= ref["terminated"]
        ref["rewards"] = rsum          # src/utils.py:126-128
        ref["terminated"] = term
        st = step_fn(st, torch.from_numpy(a1), it)
        assert_state_equal(st, ref, where=f"macro-step {it}")


def test_free_run_opponents_always_pass(env, oracle):
    """G16: single_play_step_free_run — both opponents pass, the partner plays pi.mode() (src/utils.py:205-246)."""
    from brl_amd.evaluation import masked_mode
    from brl_amd.models import make_forward_pass
    from brl_amd.utils import auto_reset, single_play_step_free_run
    n = 512
    fp = make_forward_pass("relu", "FAIR")
    actor = fp.init(3, device="cuda")
    step_fn = single_play_step_free_run(auto_reset(env.step, env.init), fp, actor)
    st = env.init(5, num_envs=n)
    ref = oracle.init_random(n, seed=5)
    for it in range(6):
        a1 = np.full(n, 3 + 5 * it, np.int32)  # 1C, 2C, ... always legal here
        a1[ref["legal_action_mask"][np.arange(n), a1] == 0] = 0
        oracle.step(ref, a1, autoreset=True, seed=5)
        rsum, term = ref["rewards"].copy(), ref["terminated"].copy()
        oracle.step(ref, np.zeros(n, np.int32), autoreset=True, seed=5)           # opponent passes
        rsum += ref["rewards"]; term |= ref["terminated"]
        obs = torch.from_numpy(ref["observation"].astype(np.float32)).cuda()
        with torch.no_grad():
            logits, _ = fp.apply(actor, obs)
        a3 = to_np(masked_mode(logits, torch.from_numpy(ref["legal_action_mask"].astype(bool)).cuda()))
        oracle.step(ref, a3, autoreset=True, seed=5)                               # partner, greedy
        rsum += ref["rewards"]; term |= ref["terminated"]
        oracle.step(ref, np.zeros(n, np.int32), autoreset=True, seed=5)           # opponent passes
        rsum += ref["rewards"]; term |= ref["terminated"]
        ref["rewards"], ref["terminated"] = rsum, term
        st = step_fn(st, torch.from_numpy(a1), it)
        assert_state_equal(st, ref, where=f"free-run macro-step {it}")


def test_lut_rotation_reinitialises_envs(dds, oracle):
    """G14 (ppo.py:525-549): swap the hash table, re-init every env; later boards come from the new table."""
    import brl_amd
    from oracle import Oracle
    k2, v2 = synthetic_lut(777, seed=9)
    env = brl_amd.BridgeBidding(lut=(dds["keys"], dds["values"]))
    st = env.init(1, num_envs=300)
    assert int(st._lut_idx.max()) < 1000
    env.set_lut((k2, v2))
    st = env.init(2, num_envs=300)
    ref = Oracle(k2, v2).init_random(300, seed=2)
    assert_state_equal(st, ref, where="after LUT rotation")
    assert int(st._lut_idx.max()) < 777


def test_graphed_update_matches_eager_update():
    """The hipGraph-captured minibatch steps must produce the same parameters as the eager one: the autograd graph
    (fused_update=False) runs the very same kernels — equal to rounding; the written-out backward + HIP Adam
    (FusedMinibatch) differs from eager by fp32 summation order only: 1e-6 after one epoch; over more steps PPO's clip
    boundaries amplify such differences for a few weights (a sample's ratio crossing 1 +- clip_eps switches its
    gradient off), so two epochs are compared on average and in their losses."""
    from brl_amd.models import make_forward_pass
    from brl_amd.update import FusedMinibatch, GraphedMinibatch, make_update_step
    from tests.test_update_cpu import CFG, fake_batch
    tb, adv, tgt = fake_batch(4, 256, seed=3)
    tb = type(tb)(*[x.cuda() for x in tb]); adv, tgt = adv.cuda(), tgt.cuda()
    fp = make_forward_pass("relu", "DeepMind")
    for epochs in (1, 2):
        outs = {}
        for mode in ("eager", "autograd-graph", "fused"):
            net = fp.init(11, device="cuda")
            cfg = dict(CFG, minibatch_size=256, update_epochs=epochs, graph_update=mode != "eager", fused_update=mode == "fused")
            rs, (total, aux) = make_update_step(cfg, fp)((net, None, None, None, 0, 5), tb, adv, tgt)
            if mode != "eager":
                assert isinstance(rs[1].get("graphed"), FusedMinibatch if mode == "fused" else GraphedMinibatch), rs[1].get("graph_error")
            assert total.shape == (epochs, 4) and all(a.shape == (epochs, 4) for a in aux)
            outs[mode] = (torch.cat([p.detach().reshape(-1) for p in net.parameters()]), total, aux)
        e = outs["eager"]
        assert torch.allclose(e[1], outs["autograd-graph"][1], atol=1e-5)
        assert torch.allclose(e[0], outs["autograd-graph"][0], atol=1e-5, rtol=1e-4)
        f = outs["fused"]
        if epochs == 1:
            assert torch.allclose(e[1], f[1], atol=1e-6) and torch.allclose(e[0], f[0], atol=1e-6, rtol=1e-5)
            for a, b in zip(e[2], f[2]):
                assert torch.allclose(a, b, atol=1e-5)
        else:
            assert torch.allclose(e[1], f[1], atol=1e-4)
            d = (e[0] - f[0]).abs()
            assert float(d.mean()) < 2e-5 and float(d.max()) < 2 * cfg["lr"]


def test_update_epoch_at_config3_size_fused_vs_eager(dds):
    """BASELINE.json configs[3]'s update at ITS size — a trajectory of 8192 x 32 (a real random rollout: observations,
    masks, actions, log-probs, rewards -> GAE), minibatch 1024, one epoch = 256 minibatch steps at the reference's lr —
    through FusedMinibatch (HIP heads / gather / Adam, graph replays) and through the eager autograd path from the same
    weights and the same permutation: the 256 logged losses agree step by step (early steps tightly, the whole epoch on
    average: PPO's clip boundaries amplify fp32 summation-order differences for single samples), and so do the weights."""
    import brl_amd
    from brl_amd.gae import gae_scan
    from brl_amd.models import make_forward_pass
    from brl_amd.train import DEFAULTS
    from brl_amd.update import FusedMinibatch, make_update_step
    env = make_env(dds, 4)
    n, T = 8192, 32
    roll = brl_amd.make_random_roll_out({"num_steps": T}, env)
    rs, traj = roll((None, None, env.init(3, num_envs=n), None, 0, 0))
    g = torch.Generator(device="cuda").manual_seed(0)
    value = torch.randn(T, n, device="cuda", generator=g) * 0.05          # (the random policy has no critic: any old values)
    traj = traj._replace(value=value)
    adv, tgt = gae_scan(env, traj.done, traj.value, traj.reward, torch.zeros(n, device="cuda"), 1.0, 0.95)
    fp = make_forward_pass("relu", "DeepMind")
    outs = {}
    for mode in ("eager", "fused"):
        net = fp.init(5, device="cuda")
        cfg = dict(DEFAULTS, lr=1e-5, minibatch_size=1024, update_epochs=1, graph_update=mode == "fused", fused_update=mode == "fused")
        rs2, (total, aux) = make_update_step(cfg, fp)((net, None, None, None, 0, 17), traj, adv, tgt)
        if mode == "fused":
            assert isinstance(rs2[1].get("graphed"), FusedMinibatch), rs2[1].get("graph_error")
        outs[mode] = (torch.cat([p.detach().reshape(-1) for p in net.parameters()]), total.reshape(-1), [a.reshape(-1) for a in aux])
    e, f = outs["eager"], outs["fused"]
    assert e[1].shape == (256,) and torch.isfinite(f[1]).all()
    assert torch.allclose(e[1][:8], f[1][:8], atol=2e-6)                   # the first steps: same arithmetic up to summation order
    assert float((e[1] - f[1]).abs().mean()) < 1e-5 and float((e[1] - f[1]).abs().max()) < 1e-3
    for k, (a, b) in enumerate(zip(e[2], f[2])):                           # value_loss, loss_actor, entropy, approx_kl, clipfrac, illegal norm
        assert float((a - b).abs().mean()) < (2e-3 if k == 4 else 2e-5), k  # (clipfrac moves in steps of 1 / 1024)
    d = (e[0] - f[0]).abs()
    moved = float((e[0] - torch.cat([p.detach().reshape(-1) for p in fp.init(5, device="cuda").parameters()])).abs().max())
    assert moved > 50 * 1e-5 and float(d.mean()) < 0.02 * moved and float(d.max()) < 0.5 * moved, (moved, float(d.mean()), float(d.max()))


def test_state_replace_board_fields(env, oracle, dds):
    """``state.replace(_hand=, _dealer=, _vul_NS=, _vul_EW=, _shuffled_players=, current_player=)`` on a fresh state
    (src/duplicate.py:120-128, wb5/utils.py:69-75, wb5/vis_pgx.py:51-56) == the oracle's explicit deal; the replaced
    hand's double-dummy tricks come from the handle's table; then the wb5 auction is played on it."""
    n = 64
    rng = np.random.default_rng(3)
    rows = rng.integers(0, 1000, n)
    hands = np.stack([oracle.key_to_hand(dds["keys"][r]) for r in rows])
    tricks = dds["tricks"][rows].reshape(n, 20)
    dealer = rng.integers(0, 4, n).astype(np.int32)
    vns, vew = rng.integers(0, 2, n).astype(np.int32), rng.integers(0, 2, n).astype(np.int32)
    perms = np.array([[0, 3, 1, 2], [2, 0, 3, 1], [1, 2, 0, 3], [3, 1, 2, 0]], np.int32)
    shuf = perms[rng.integers(0, 4, n)]
    st = env.init(99, num_envs=n)
    cur = shuf[np.arange(n), dealer]
    st = st.replace(_hand=torch.from_numpy(hands), _dealer=torch.from_numpy(dealer), _vul_NS=torch.from_numpy(vns).bool(),
                    _vul_EW=torch.from_numpy(vew).bool(), _shuffled_players=torch.from_numpy(shuf),
                    current_player=torch.from_numpy(cur))
    ref = oracle.init_explicit(hands, dealer, vns, vew, shuf, tricks)
    skip = {"lut_idx", "board_ctr"}
    fields = set(__import__("tests.gpu_util", fromlist=["FIELD_MAP"]).FIELD_MAP) - skip
    assert_state_equal(st, ref, fields=fields, where="replace(board fields)")
    assert np.array_equal(to_np(st._lut_idx), rows)
    for i, a in enumerate([0, 9, 11, 20, 1, 0, 22, 1, 2, 0, 0, 28, 0, 0, 0]):   # wb5/utils.py:61-64 (+ the closing pass)
        st = env.step(st, torch.full((n,), a, dtype=torch.int32))
        oracle.step(ref, np.full(n, a, np.int32))
        assert_state_equal(st, ref, fields=fields, where=f"replace + call {i}")
    assert ref["terminated"].all() and np.abs(ref["rewards"]).max() > 0
    with pytest.raises(ValueError):
        env.init(1, num_envs=4).replace(_dealer=2, current_player=torch.tensor([9, 9, 9, 9]))
    with pytest.raises(NotImplementedError):
        env.init(1, num_envs=4).replace(_turn=3)
    # a deal that is not in the table: zero tricks, row -1
    other = np.stack([np.random.default_rng(i).permutation(52) for i in range(4)])
    s2 = env.init(1, num_envs=4).replace(_hand=torch.from_numpy(other))
    assert (to_np(s2._dds_tricks) == 0).all() and (to_np(s2._lut_idx) == -1).all()
    assert np.array_equal(np.sort(to_np(s2._hand).reshape(4, 4, 13), 2), np.sort(other.reshape(4, 4, 13), 2))


@pytest.mark.parametrize("graph", [False, True])
def test_update_step_matches_numpy_restatement_gpu(graph):
    """§8f-1 on the device: one full PPO minibatch step at minibatch 1024 (forward, _loss_fn, backward, global-norm
    clip, Adam) vs the float64 numpy restatement, eager and hipGraph-replayed (tolerances in tests/test_update_cpu.py)."""
    from tests.test_update_cpu import check_update_against_numpy
    check_update_against_numpy("cuda", graph=graph, T=4, N=256)


def test_ppo_loss_on_device_matches_numpy_at_minibatch_1024():
    from brl_amd.models import make_forward_pass
    from brl_amd.roll_out import Transition
    from brl_amd.update import ppo_loss
    from tests.test_update_cpu import CFG, fake_batch, numpy_loss
    tb, adv, tgt = fake_batch(8, 128, seed=5)
    flat = Transition(*[x.reshape((1024,) + x.shape[2:]).cuda() for x in tb])
    fp = make_forward_pass("relu", "DeepMind")
    net = fp.init(2, device="cuda")
    for coef in (0.0, 0.25):
        cfg = dict(CFG, illegal_action_l2norm_coef=coef)
        with torch.no_grad():
            logits, value = fp.apply(net, flat.obs.float())
            total, aux = ppo_loss(cfg, logits, value, flat, adv.reshape(-1).cuda(), tgt.reshape(-1).cuda())
        cpu = Transition(*[x.cpu() for x in flat])
        want = numpy_loss(cfg, logits.cpu(), value.cpu(), cpu, adv.reshape(-1), tgt.reshape(-1))
        assert abs(float(total) - want[0]) < 1e-5                              # fp32 reductions vs fp64
        for k in range(3):
            assert abs(float(aux[k]) - want[k + 1]) < 1e-5
        assert abs(float(aux[5]) - want[4]) < 1e-4 * want[4]                   # illegal-action norm (SVD-free when coef == 0)


@pytest.mark.parametrize("masked,vclip,B", [(True, True, 1024), (False, True, 1024), (True, False, 333)])
def test_fused_ppo_loss_kernel_matches_numpy(masked, vclip, B):
    """brl_ppo_loss / brl_spectral_half_norm (one launch each) vs the float64 restatement tests/ppo_numpy.head_loss:
    the five statistics, the illegal-action norm and the gradient w.r.t. logits / value that autograd would produce.
    Tolerances: fp32 exp / log and 1024-term sums vs fp64 — 2e-6 absolute on per-sample gradients of size <= 1e-3."""
    from brl_amd.roll_out import Transition
    from brl_amd.update import ppo_loss, ppo_loss_fused
    from tests.ppo_numpy import head_loss
    from tests.test_update_cpu import CFG, fake_batch
    T = 1
    tb, adv, tgt = fake_batch(T, B, seed=17)
    flat = Transition(*[x.reshape((B,) + x.shape[2:]).cuda() for x in tb])
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(B, 38, generator=g) * 1.5).cuda().requires_grad_(True)
    value = (torch.randn(B, generator=g) * 0.2).cuda().requires_grad_(True)
    # a stored log_prob near the current one, so that ratios straddle the clip range
    with torch.no_grad():
        lsm = torch.log_softmax(torch.where(flat.legal_action_mask, logits, torch.full_like(logits, -1e30)) if masked else logits, -1)
        old_lp = lsm.gather(1, flat.action.long()[:, None])[:, 0] + 0.3 * torch.randn(B, generator=g).cuda()
    flat = flat._replace(log_prob=old_lp)
    cfg = dict(CFG, actor_illegal_action_mask=masked, value_clipping=vclip)
    total, aux, (dl, dv) = ppo_loss_fused(cfg, logits, value, flat, adv.reshape(-1).cuda(), tgt.reshape(-1).cuda())
    want = head_loss(cfg, logits.detach().double().cpu().numpy(), value.detach().double().cpu().numpy(),
                     to_np(flat.legal_action_mask), to_np(flat.action).astype(np.int64), flat.value.double().cpu().numpy(),
                     old_lp.double().cpu().numpy(), adv.reshape(-1).double().numpy(), tgt.reshape(-1).double().numpy())
    assert abs(float(total) - want[0]) < 2e-5
    for k in range(5):
        assert abs(float(aux[k]) - want[1][k]) < 2e-5, k
    assert 0.05 < float(aux[4]) < 0.95                                       # both clip branches exercised
    assert np.abs(to_np(dl).astype(np.float64) - want[2]).max() < 2e-6 and np.abs(want[2]).max() > 1e-5
    assert np.abs(to_np(dv).astype(np.float64) - want[3]).max() < 2e-6
    p = torch.softmax(logits.detach().double(), -1) * (~flat.legal_action_mask)
    sv = float(torch.linalg.matrix_norm(p, ord=2)) / 2
    assert abs(float(aux[5]) - sv) < 1e-4 * sv                               # SVD-free norm: 1e-4 relative
    # and against torch autograd on the unfused loss (same device, fp32)
    t2, aux2 = ppo_loss(cfg, logits, value, flat, adv.reshape(-1).cuda(), tgt.reshape(-1).cuda())
    t2.backward()
    assert torch.allclose(logits.grad, dl, atol=2e-6) and torch.allclose(value.grad, dv, atol=2e-6)
    assert abs(float(t2) - float(total)) < 2e-5


@pytest.mark.parametrize("fused", [False, True])
def test_graphed_update_built_after_eager_steps_keeps_adam_state(fused):
    """A hipGraph minibatch step captured AFTER eager steps (optimizer moments / step counts already live) must continue
    from that state: same parameters as staying eager (autograd graph: to rounding; FusedMinibatch, which adopts the
    moments into its flat buffers: on average — see test_graphed_update_matches_eager_update for why not element-wise)."""
    from brl_amd.models import make_forward_pass
    from brl_amd.update import make_update_step
    from tests.test_update_cpu import CFG, fake_batch
    fp = make_forward_pass("relu", "DeepMind")
    outs = []
    for late_graph in (False, True):
        net = fp.init(11, device="cuda")
        rs = (net, None, None, None, 0, 5)
        for it in range(3):
            tb, adv, tgt = fake_batch(4, 256, seed=20 + it)
            tb = type(tb)(*[x.cuda() for x in tb])
            cfg = dict(CFG, minibatch_size=256, update_epochs=1, graph_update=(late_graph and it >= 1), fused_update=fused)
            rs, _ = make_update_step(cfg, fp)(rs, tb, adv.cuda(), tgt.cuda())
        if late_graph:
            assert rs[1].get("graphed"), rs[1].get("graph_error")
        outs.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
        steps = {int(st["step"]) for st in rs[1]["opt"].state.values()}
        assert steps == {12}   # 3 updates x 4 minibatches, none lost to the capture's warm-up
    if fused:
        d = (outs[0] - outs[1]).abs()
        assert float(d.mean()) < 2e-5 and float(d.max()) < 2 * CFG["lr"]
    else:
        assert torch.allclose(outs[0], outs[1], atol=1e-5, rtol=1e-4)


def _fused_rank(rank, world, port, out_dir, backend="gloo", mode="sharded", in_graph=None, model="DeepMind", per_layer=True):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from brl_amd.models import make_forward_pass
    from brl_amd.update import FusedStep, make_update_step
    from tests.test_update_cpu import CFG, fake_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":                                           # RCCL: one GPU per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU of the box
    fp = make_forward_pass("relu", model)
    net = fp.init(11, device="cuda")
    tb, adv, tgt = fake_batch(4, 256, seed=3)                       # the SAME shard on both ranks: mean gradient == own gradient
    tb = type(tb)(*[x.cuda() for x in tb])
    cfg = dict(CFG, minibatch_size=256, update_epochs=2, grad_allreduce=mode, force_collectives=True, collective_in_graph=in_graph,
               per_layer_dw=per_layer)   # (True: flat by the sharded form's launches — the two forms must then agree bit for bit)
    rs, (total, _) = make_update_step(cfg, fp)((net, None, None, None, 0, 5), tb, adv.cuda(), tgt.cuda())
    fm = rs[1].get("graphed")
    assert isinstance(fm, FusedStep) and fm.world == world, rs[1].get("graph_error")
    assert fm.allreduce_mode == mode and fm.in_graph == (backend == "nccl" if in_graph is None else in_graph)
    if not fm.in_graph:   # graphs of kernel groups between eager collectives: flat 1, sharded (buckets) + 1 + (buckets)
        nb = len(net.body) if model.startswith("DeepMind") else 1
        assert len([x for x in fm.segs if x[0] == "c"]) == (1 if mode == "flat" else 2 * nb + 1)
    fm.gather_optimizer_state()     # (sharded: the moments of the other ranks' slices)
    flat = lambda ts: torch.cat([t.detach().reshape(-1) for t in ts]).cpu()   # noqa: E731
    st = rs[1]["opt"].state
    torch.save((flat(net.parameters()), total.cpu(), flat([st[q]["exp_avg"] for q in net.parameters()]),
                flat([st[q]["exp_avg_sq"] for q in net.parameters()])), os.path.join(out_dir, f"{mode}{'' if per_layer else '_default'}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _single_process_reference(model="DeepMind"):
    from brl_amd.models import make_forward_pass
    from brl_amd.update import make_update_step
    from tests.test_update_cpu import CFG, fake_batch
    fp = make_forward_pass("relu", model)
    net = fp.init(11, device="cuda")
    tb, adv, tgt = fake_batch(4, 256, seed=3)
    tb = type(tb)(*[x.cuda() for x in tb])
    rs, (total, _) = make_update_step(dict(CFG, minibatch_size=256, update_epochs=2), fp)((net, None, None, None, 0, 5), tb, adv.cuda(), tgt.cuda())
    st = rs[1]["opt"].state
    flat = lambda ts: torch.cat([t.detach().reshape(-1) for t in ts]).cpu()   # noqa: E731
    return flat(net.parameters()), total.cpu(), flat([st[q]["exp_avg"] for q in net.parameters()])


def _check_two_rank_step(tmp_path, backend, world=2, in_graph=None, model="DeepMind"):
    """both forms of the multi-rank step: every rank ends with the same parameters, "sharded" == "flat" BIT FOR BIT (parameters
    and — after gather_optimizer_state — both Adam moments), and both equal the single-process step up to the fp32 order of the
    norm's partial sums"""
    import socket
    import torch.multiprocessing as mp
    res = {}
    for mode in ("flat", "sharded"):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        mp.start_processes(_fused_rank, args=(world, port, str(tmp_path), backend, mode, in_graph, model), nprocs=world, join=True,
                           start_method="spawn")
        res[mode] = [torch.load(tmp_path / f"{mode}{r}.pt") for r in range(world)]
        for r in range(1, world):
            for a, b in zip(res[mode][0], res[mode][r]):
                assert torch.equal(a, b), f"{mode}: rank {r} differs from rank 0"
    for a, b in zip(res["flat"][0], res["sharded"][0]):
        assert torch.equal(a, b)                       # parameters, losses, exp_avg, exp_avg_sq
    # against the single-process step (other launches for the weight gradients, another order of the norm's partial sums): the
    # first epoch's losses to 2e-6; over the 8 steps two correct fp32 implementations drift apart — a pre-activation of ~1e-9
    # puts a ReLU gate on different sides (scripts/relu_kink_probe.py: the six-layer net hits two such units at step 1: 1.6 % in one
    # bias gradient) and Adam turns that into +- lr moves — so the parameters are compared with a bound that a real error (a
    # misplaced slice, a missing 1 / world) would exceed by orders of magnitude
    single, total, m1 = _single_process_reference(model)
    got = res["sharded"][0]
    from tests.test_update_cpu import CFG as _CFG
    d = (single - got[0]).abs()
    assert torch.allclose(total[0], got[1][0], atol=2e-6) and torch.allclose(total, got[1], atol=1e-3)
    assert float(d.mean()) < 2e-4 and float(d.max()) < 8 * _CFG["lr"], (float(d.mean()), float(d.max()))
    # ... and the DEFAULT flat form (the single-rank launches: batched weight gradients): the same up to fp32 summation order
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.start_processes(_fused_rank, args=(world, port, str(tmp_path), backend, "flat", in_graph, model, False), nprocs=world, join=True,
                       start_method="spawn")
    dflt = [torch.load(tmp_path / f"flat_default{r}.pt") for r in range(world)]
    for r in range(1, world):
        assert torch.equal(dflt[0][0], dflt[r][0])
    d = (single - dflt[0][0]).abs()
    assert float(d.mean()) < 2e-4 and float(d.max()) < 8 * _CFG["lr"] and torch.allclose(total, dflt[0][1], atol=1e-3)


def test_fused_update_with_gradient_collectives_two_ranks(tmp_path):
    """FusedMinibatch under a process group — "flat" (the default: ONE all-reduce, replicated clip + Adam) and "sharded"
    (reduce-scatter per layer bucket behind the backward pass, clip + Adam on the rank's slices, all-gather of the parameters): two
    gloo ranks on this box's GPU (graphs of kernel groups between eager collectives; RCCL captures them into the step's graph)."""
    _check_two_rank_step(tmp_path, "gloo")


def test_fused_fair_update_with_gradient_collectives_two_ranks(tmp_path):
    """FusedFair under a process group (one bucket: all-reduce + replicated sweep, or reduce-scatter + the rank's slice + all-gather):
    two gloo ranks, both forms bit-identical and equal to the single-process step; then the same with REAL RCCL nodes inside the
    graph at world 1."""
    _check_two_rank_step(tmp_path, "gloo", model="FAIR")
    _check_two_rank_step(tmp_path, "nccl", world=1, model="FAIR")


def test_fused_update_with_gradient_collectives_two_ranks_rccl(tmp_path):
    """The same over RCCL ("nccl"), one GPU per rank: the collectives are nodes of the eight-step hipGraph and really cross xGMI
    ("sharded": the reduce-scatters overlap the backward pass, the all-gathers the next forward pass).  SKIPPED below 2 GPUs."""
    _two_gpus_or_skip()
    _check_two_rank_step(tmp_path, "nccl")


def test_fused_update_collectives_inside_the_graph_rccl_world_1(tmp_path):
    """The multi-rank program with REAL RCCL collectives captured inside the step's hipGraph, on the one GPU of this box
    (init_process_group("nccl", world_size=1), force_collectives): reduce_scatter_tensor / all_gather_into_tensor / all_reduce nodes
    record and replay (profiles/r05/r05a_rccl_capture_probe.txt), "sharded" == "flat" bit for bit, both == the single-rank step up to
    the order of the norm's partial sums.  What a node run adds is peers, not code."""
    _check_two_rank_step(tmp_path, "nccl", world=1)


def test_fused_update_collectives_of_the_six_layer_mlp(tmp_path):
    """wb5/models.py's six-layer MLP ("DeepMind_6": six buckets): two gloo ranks, both forms bit-identical and equal to the single-process
    step; then the same with real RCCL nodes inside the graph at world 1."""
    _check_two_rank_step(tmp_path, "gloo", model="DeepMind_6")
    _check_two_rank_step(tmp_path, "nccl", world=1, model="DeepMind_6")


def test_fused_update_sharded_geometry_of_eight_ranks_on_one_gpu(tmp_path):
    """configs[4]'s geometry (world = 8: five buckets x eight slices, 1024 norm partials) executed by ONE process that plays the
    eight ranks in turn through the C-ABI: reduce-scatter / all-gather done by hand on host-visible copies.  The eight sharded
    sweeps together == one replicated sweep (rank_lo = 0, rank_hi = 8) bit for bit == torch.optim.Adam + clip_grad_norm_ to fp32
    rounding."""
    import ctypes as C
    from brl_amd import _capi
    L, dev = _capi.lib(), torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(8)
    W, H, nl = 8, 1024, 4
    hid = (nl - 1) * H * H
    tail = 480 * H + 39 * H + nl * H + 39
    tail_pad = (tail + 4 * W - 1) // (4 * W) * (4 * W)
    n = hid + tail_pad
    geom = _capi.ShardGeom()
    geom.nbuckets, geom.world, geom.nsub = nl, W, 1024 // (W * nl)
    for b in range(nl - 1):
        geom.off[b], geom.len[b] = b * H * H, H * H // W
    geom.off[nl - 1], geom.len[nl - 1] = hid, tail_pad // W
    npart = W * nl * geom.nsub
    p0 = torch.randn(n, device=dev, generator=g) * 0.05
    grads = [torch.randn(n, device=dev, generator=g) * (2e-4 if it == 1 else 1e-2) for it in range(3)]   # one step below the clip threshold
    for t in [p0] + grads:
        t[hid + tail:] = 0                                   # the buffers' zero padding
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, eps=1e-5)

    def run(sharded):
        p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        step, norm = torch.zeros((), device=dev), torch.zeros(1, device=dev)
        idx = torch.zeros(1, dtype=torch.int32, device=dev)
        norms = []
        for grad in grads:
            gsum = grad * W                                     # what a SUM reduction of eight equal gradients leaves
            part = torch.full((npart,), float("nan"), device=dev)
            if sharded:
                step0 = step.clone()
                for r in range(W):                              # every rank: partials of its own slices (its own step counter)
                    step.copy_(step0)
                    _capi.check(L.brl_adam_shard_norm(0, gsum.data_ptr(), C.byref(geom), r, r + 1, 1.0 / W, part.data_ptr(), step.data_ptr(),
                                                      idx.data_ptr() if r == 0 else None, s))
                for r in range(W):                              # (the all-gather of the partials has happened: one shared array)
                    _capi.check(L.brl_adam_shard_apply(0, p.data_ptr(), gsum.data_ptr(), m.data_ptr(), v.data_ptr(), C.byref(geom), r, r + 1,
                                                       part.data_ptr(), step.data_ptr(), 1e-3, None, 0.9, 0.999, 1e-5, 0.5, 1.0 / W,
                                                       norm.data_ptr(), None, 0, s))
            else:
                _capi.check(L.brl_adam_shard_norm(0, gsum.data_ptr(), C.byref(geom), 0, W, 1.0 / W, part.data_ptr(), step.data_ptr(),
                                                  idx.data_ptr(), s))
                _capi.check(L.brl_adam_shard_apply(0, p.data_ptr(), gsum.data_ptr(), m.data_ptr(), v.data_ptr(), C.byref(geom), 0, W,
                                                   part.data_ptr(), step.data_ptr(), 1e-3, None, 0.9, 0.999, 1e-5, 0.5, 1.0 / W,
                                                   norm.data_ptr(), None, 0, s))
            assert not torch.isnan(part).any()
            norms.append(float(norm[0]))
        assert float(step) == 3.0 and int(idx) == 3
        return p, m, v, norms
    a, b = run(True), run(False)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert a[3] == b[3]
    for it, grad in enumerate(grads):
        ref.grad = grad.clone()
        want_norm = float(torch.nn.utils.clip_grad_norm_([ref], 0.5))
        opt.step()
        assert abs(a[3][it] - want_norm) < 1e-5 * want_norm
    assert torch.allclose(a[0], ref.detach(), atol=2e-6), float((a[0] - ref.detach()).abs().max())


@pytest.mark.parametrize("chain", [True, False])
@pytest.mark.parametrize("activation,rscale,masked", [("relu", False, True), ("tanh", False, True), ("relu", True, True), ("relu", False, False)])
def test_fused_fair_step_matches_numpy_restatement(activation, rscale, masked, chain):
    """FusedFair (src/models.py:34-69 written out: forward, `_loss_fn`, backward on flat buffers, clip + Adam) — ONE minibatch step
    at minibatch 1024 vs the float64 numpy restatement (tests/ppo_numpy.py::fair_loss_and_grads, itself checked against autograd in
    float64 on the CPU): every gradient before the sweep, the losses, the pre-clip norm, every parameter after the first Adam step.
    chain: forward + loss + backward chain as ONE launch (brl_fair_chain, the default) / as ~55 launches (library products +
    brl_mlp_gemm + elementwise kernels)."""
    from brl_amd.models import make_forward_pass
    from brl_amd.roll_out import Transition
    from brl_amd.update import FusedFair, make_update_step
    from tests.ppo_numpy import adam_first_step, fair_loss_and_grads, fair_params_of
    from tests.test_update_cpu import CFG, fake_batch
    tb, adv, tgt = fake_batch(4, 256, seed=2)
    B = 1024
    cfg = dict(CFG, minibatch_size=B, update_epochs=1, lr=1e-3, reward_scaling=rscale, actor_illegal_action_mask=masked, fair_chain=chain)
    fp = make_forward_pass(activation, "FAIR")
    net = fp.init(4, device="cuda")
    if activation == "tanh":
        # hk.Linear's biases start at zero: a wrong bias index in the chain's forward would change nothing — perturb EVERY parameter,
        # the biases by ~0.1 (test_fair_forward does the same for the <false> form).  tanh cases only: the index arithmetic does not
        # depend on the activation, and under ReLU a perturbed net puts a pre-activation or two of the 1024 x 11 x 200 within fp32
        # rounding of 0 — its gate falls on the other side than in float64 and one sample's share (~1e-5) appears in a few gradient
        # entries, which Adam's first step (lr g / (|g| + eps)) turns into lr-sized differences (profiles/r05/r05n_relu_kink_probe.txt)
        with torch.no_grad():
            gen = torch.Generator(device="cuda").manual_seed(11)
            for q in net.parameters():
                q.add_(torch.randn(q.shape, device="cuda", generator=gen) * (0.1 if q.dim() == 1 else 0.01))
    P0 = fair_params_of(net)
    flat = Transition(*[x.reshape((B,) + x.shape[2:]) for x in tb])
    gae64 = adv.reshape(-1).double().numpy()
    if rscale:                                                  # src/update.py:31-44 (jnp std: ddof = 0)
        gae64 = (gae64 - gae64.mean()) / (gae64.std() + 1e-8)
    want_total, want_aux, G = fair_loss_and_grads(cfg, P0, flat.obs.numpy(), flat.legal_action_mask.numpy(), flat.action.numpy().astype(np.int64),
                                                  flat.value.double().numpy(), flat.log_prob.double().numpy(), gae64,
                                                  tgt.reshape(-1).double().numpy(), activation=activation)
    P1, gn = adam_first_step(cfg, P0, G)
    rs, (total, aux) = make_update_step(cfg, fp)((net, None, None, None, 0, 9), Transition(*[x.cuda() for x in tb]), adv.cuda(), tgt.cuda())
    fm = rs[1].get("graphed")
    assert isinstance(fm, FusedFair) and fm.chain == chain, rs[1].get("graph_error")
    assert abs(float(total[0, 0]) - want_total) < 2e-5
    for k in range(5):
        assert abs(float(aux[k][0, 0]) - want_aux[k]) < 2e-5, k
    assert abs(float(fm.norm[0]) - gn) < 1e-4 * gn
    lins = list(net.l) + [net.actor, net.critic]
    scale = max(np.abs(gw).max() for gw, _ in G)
    for lin, (gw, gb) in zip(lins, G):     # the gradients the step left in the flat buffer (the sweep does not change them)
        assert np.abs(lin.weight.grad.cpu().double().numpy() - gw).max() < 2e-5 * scale + 1e-9
        assert np.abs(lin.bias.grad.cpu().double().numpy() - gb).max() < 2e-5 * scale + 1e-9
    got = fair_params_of(net)
    worst = max(max(np.abs(a - c).max(), np.abs(b - d).max()) for (a, b), (c, d) in zip(got, P1))
    moved = max(np.abs(a - c).max() for (a, _), (c, _) in zip(P0, P1))
    assert worst < 0.02 * cfg["lr"] and moved > 0.5 * cfg["lr"], (worst, moved, gn)


@pytest.mark.parametrize("activation,n", [("relu", 16), ("relu", 1000), ("tanh", 333), ("relu", 8192)])
def test_fair_forward_matches_float64(activation, n):
    """brl_fair_forward (`actor(x), critic(x)` of the FAIR network, src/models.py:34-69, as one launch) — through the module's own
    inference path (no autograd: ActorCritic._fair_forward) against the same module in float64 on the host, against the launch-by-launch
    torch forward on the GPU, and against the oracle shim's float64 restatement of the entry point; with the heads as two separate
    parameters (a fresh module) and as one [39, 200] block (what FusedFair's flat buffer makes of them)."""
    import ctypes as C
    import oracle
    from brl_amd import _capi
    from brl_amd.models import make_forward_pass
    from oracle.binding import shim_path
    fp = make_forward_pass(activation, "FAIR")
    net = fp.init(3, device="cuda")
    with torch.no_grad():
        for q in net.parameters():
            q.add_(torch.randn_like(q) * 0.05)          # (hk.Linear's zero biases would hide a bias mix-up)
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.rand((n, 480), device="cuda", generator=g) < 0.12).float()
    with torch.no_grad():
        lg, v = net(x)
        os.environ["BRL_FAIR_FORWARD"] = "0"
        try:
            lg_t, v_t = net(x)
        finally:
            del os.environ["BRL_FAIR_FORWARD"]
        ref = fp.init(3, device="cpu").double()
        ref.load_state_dict({k: t.double().cpu() for k, t in net.state_dict().items()})
        lg64, v64 = ref(x.double().cpu())
    assert lg.shape == (n, 38) and v.shape == (n,)
    scale = max(1.0, float(lg64.abs().max()))
    assert float((lg.double().cpu() - lg64).abs().max()) < 2e-5 * scale and float((v.double().cpu() - v64).abs().max()) < 2e-5 * scale
    assert float((lg - lg_t).abs().max()) < 2e-5 * scale and float((v - v_t).abs().max()) < 2e-5 * scale
    # the heads as ONE block behind each other (FusedFair's flat layout): the zero-copy branch gives the same bits
    hw = torch.cat([net.actor.weight.detach(), net.critic.weight.detach()], 0).contiguous()
    hb = torch.cat([net.actor.bias.detach(), net.critic.bias.detach()], 0).contiguous()
    net.actor.weight.data, net.critic.weight.data, net.actor.bias.data, net.critic.bias.data = hw[:38], hw[38:], hb[:38], hb[38:]
    with torch.no_grad():
        lg2, v2 = net(x)
    assert torch.equal(lg, lg2) and torch.equal(v, v2)
    # the oracle shim's restatement of the same entry point (float64 accumulation), a few rows
    oracle.build()
    shim = C.CDLL(shim_path())
    m = min(n, 24)
    fnet = _capi.FairNet()
    keep = []
    for l, lin in enumerate(net.l):
        w, b = lin.weight.detach().cpu().contiguous().numpy(), lin.bias.detach().cpu().contiguous().numpy()
        keep += [w, b]
        fnet.w[l], fnet.b[l] = w.ctypes.data, b.ctypes.data
    hwn, hbn, xn = hw.cpu().numpy(), hb.cpu().numpy(), x[:m].cpu().contiguous().numpy()
    fnet.head_w, fnet.head_b = hwn.ctypes.data, hbn.ctypes.data
    lo, vo = np.zeros((m, 38), np.float32), np.zeros(m, np.float32)
    shim.brl_fair_forward.argtypes = [C.c_int, C.POINTER(_capi.FairNet), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert shim.brl_fair_forward(0, fnet, xn.ctypes.data, m, 0 if activation == "relu" else 1, lo.ctypes.data, vo.ctypes.data, None) == 0
    assert np.abs(lo - lg[:m].cpu().numpy()).max() < 2e-5 * scale and np.abs(vo - v[:m].cpu().numpy()).max() < 2e-5 * scale


@pytest.mark.parametrize("model,activation", [("FAIR", "relu"), ("DeepMind", "tanh")])
def test_graphed_rollout_follows_the_parameters_after_updates(model, activation, tmp_path):
    """The architectures without an inference snapshot (FAIR; tanh) run `module(x)` inside the captured rollout, and the fused
    update re-points the module's parameters into its flat buffers: after every update the replayed rollout must read the LIVE
    parameters — scan step 0's values recomputed eagerly from traj.obs[0] with the loop's current parameters, four iterations."""
    from brl_amd.train import train
    cfg = dict(num_envs=1024, num_steps=8, minibatch_size=1024, update_epochs=2, total_timesteps=1024 * 8 * 4, graph_rollout=True,
               evaluate=False, save_model=False, log_path=str(tmp_path), exp_name="stale", actor_model_type=model,
               actor_activation=activation, lr=1e-2)
    seen = []

    def on_rollout(i, rs, traj, roll_out):
        with torch.no_grad():
            _, v = rs[0](traj.obs[0].float())
        seen.append((float((v - traj.value[0]).abs().max()), float(v.abs().max())))

    train(cfg, log=lambda s: None, on_rollout=on_rollout)
    assert len(seen) == 4 and all(d <= 1e-6 * max(1.0, m) for d, m in seen), seen
    assert abs(seen[0][1] - seen[-1][1]) > 1e-4      # (the parameters did move)


def test_graphed_rollout_on_planes_follows_the_parameters_after_updates(tmp_path):
    """The DeepMind / ReLU policy at 4096 tables: the captured rollout's forwards run on brl_linear_x3p with the weights' bf16 planes held by
    the inference snapshot; every iteration's `refresh` re-splits the UPDATED weights into the same plane tensors (their addresses sit in
    the graphs).  Scan step 0's values recomputed eagerly from traj.obs[0] with the loop's current parameters, three iterations, lr large
    enough that stale planes would show."""
    from brl_amd.train import train
    cfg = dict(num_envs=4096, num_steps=4, minibatch_size=1024, update_epochs=2, total_timesteps=4096 * 4 * 3, graph_rollout=True,
               evaluate=False, save_model=False, log_path=str(tmp_path), exp_name="planes", actor_model_type="DeepMind",
               actor_activation="relu", lr=1e-2)
    seen = []

    def on_rollout(i, rs, traj, roll_out):
        eng = roll_out.engine
        assert eng.xin.dtype == torch.bfloat16 and eng.snap_actor.wp is not None
        with torch.no_grad():
            _, v = rs[0](traj.obs[0].float())
        seen.append((float((v - traj.value[0]).abs().max()), float(v.abs().max())))

    train(cfg, log=lambda s: None, on_rollout=on_rollout)
    assert len(seen) == 3 and all(d <= 2e-4 * max(1.0, m) for d, m in seen), seen
    assert abs(seen[0][1] - seen[-1][1]) > 1e-4      # (the parameters did move)


@pytest.mark.parametrize("variant", ["relu", "tanh", "reward_scaling", "unmasked", "launches", "library_gemms", "illegal_coef"])
def test_fused_fair_update_matches_eager(variant):
    """FusedFair vs the eager autograd path from the same start: ONE update of one epoch x 4 minibatches (the single step is checked
    against float64 above; over more steps Adam turns the rounding differences of near-zero gradients into +- lr moves) — parameters
    (mean 1e-5, max 2 lr: the bound of test_fused_update_variants_match_eager), the logged rows of every step, the step counters;
    then a second update on the fused path alone: finite, counters at 8."""
    from brl_amd.models import make_forward_pass
    from brl_amd.update import FusedFair, make_optimizer, make_update_step
    from tests.test_update_cpu import CFG, fake_batch
    fp = make_forward_pass("tanh" if variant == "tanh" else "relu", "FAIR")
    cfg0 = dict(CFG, minibatch_size=256, update_epochs=1, reward_scaling=variant == "reward_scaling",
                illegal_action_l2norm_coef=0.5 if variant == "illegal_coef" else 0.0,     # src/update.py:146-152 (launch by launch)
                actor_illegal_action_mask=variant != "unmasked", own_gemm=variant != "library_gemms",
                fair_chain=variant not in ("launches", "library_gemms"))    # (the default: brl_fair_chain; else launch by launch)
    tb, adv, tgt = fake_batch(4, 256, seed=40)
    tb = type(tb)(*[x.cuda() for x in tb])
    outs = []
    for fused in (False, True):
        net = fp.init(7, device="cuda")
        cfg = dict(cfg0, graph_update=fused, fused_update=fused)
        rs = (net, make_optimizer(cfg, net), None, None, 0, 5)
        rs, (total, aux) = make_update_step(cfg, fp)(rs, tb, adv.cuda(), tgt.cuda())
        if fused:
            assert isinstance(rs[1].get("graphed"), FusedFair), rs[1].get("graph_error")
        assert {int(st["step"]) for st in rs[1]["opt"].state.values()} == {4}
        outs.append((torch.cat([p.detach().reshape(-1) for p in net.parameters()]), total.clone(), torch.stack(list(aux)).clone()))
    d = (outs[0][0] - outs[1][0]).abs()
    assert float(d.mean()) < 1e-5 and float(d.max()) < 2 * CFG["lr"], (float(d.mean()), float(d.max()))
    assert torch.allclose(outs[0][1], outs[1][1], atol=2e-4) and torch.allclose(outs[0][2][:5], outs[1][2][:5], atol=2e-4)
    assert torch.allclose(outs[0][2][5], outs[1][2][5], rtol=2e-3, atol=1e-6)     # the illegal-action norm (no SVD on the fused path)
    rs, (total, _) = make_update_step(cfg, fp)(rs, tb, adv.cuda(), tgt.cuda())
    assert bool(torch.isfinite(total).all()) and {int(st["step"]) for st in rs[1]["opt"].state.values()} == {8}


@pytest.mark.parametrize("variant", ["DeepMind_6", "anneal_lr", "tanh", "reward_scaling", "illegal_coef", "library_gemms", "own_gemm_fwd", "dw_bf16x3",
                                     "dw_bf16x3_6"])
def test_fused_update_variants_match_eager(variant):
    """FusedMinibatch on the 6-layer MLP of wb5/models.py, under ppo.py:186-192's linear learning-rate schedule (the
    rate lives in device memory, so the captured Adam launch follows it), with the tanh activation (src/models.py:16) and
    with reward_scaling (src/update.py:31-44,118): two updates of one epoch vs the eager autograd path."""
    from brl_amd.models import make_forward_pass
    from brl_amd.update import FusedMinibatch, make_optimizer, make_update_step
    from tests.test_update_cpu import CFG, fake_batch
    fp = make_forward_pass("tanh" if variant == "tanh" else "relu", "DeepMind_6" if variant in ("DeepMind_6", "dw_bf16x3_6") else "DeepMind")
    cfg0 = dict(CFG, minibatch_size=256, update_epochs=1, num_minibatches=4, num_updates=4, anneal_lr=variant == "anneal_lr",
                # the default: every weight gradient as ONE brl_mlp_gemm_x3_group launch; "library": torch.bmm + torch.mm
                dw_gemm="library" if variant in ("library_gemms", "DeepMind_6", "tanh") else None,
                reward_scaling=variant == "reward_scaling",
                illegal_action_l2norm_coef=0.5 if variant == "illegal_coef" else 0.0,   # src/update.py:146-152
                # the step's switches: every product with the library + brl_act_bwd_colsum[_heads_dw] / the forward layers on brl_mlp_gemm too
                own_gemm=variant != "library_gemms", own_gemm_fwd=variant == "own_gemm_fwd")
    outs = []
    for fused in (False, True):
        net = fp.init(7, device="cuda")
        cfg = dict(cfg0, graph_update=fused)
        rs = (net, make_optimizer(cfg, net), None, None, 0, 5)
        for it in range(2):
            tb, adv, tgt = fake_batch(4, 256, seed=40 + it)
            rs, _ = make_update_step(cfg, fp)(rs, type(tb)(*[x.cuda() for x in tb]), adv.cuda(), tgt.cuda())
        if fused:
            assert isinstance(rs[1].get("graphed"), FusedMinibatch), rs[1].get("graph_error")
            assert rs[1]["graphed"].dw_x3 == (variant not in ("library_gemms", "DeepMind_6", "tanh"))
        outs.append((torch.cat([p.detach().reshape(-1) for p in net.parameters()]), rs[1]["opt"].param_groups[0]["lr"]))
    assert abs(outs[0][1] - outs[1][1]) < 1e-12 and (variant != "anneal_lr" or abs(outs[0][1] - 0.5 * CFG["lr"]) < 1e-12)
    d = (outs[0][0] - outs[1][0]).abs()
    assert float(d.mean()) < 1e-5 and float(d.max()) < 2 * CFG["lr"], (float(d.mean()), float(d.max()))


def test_fused_update_follows_a_reloaded_optimizer_state():
    """opt.load_state_dict between two updates replaces the moment tensors FusedMinibatch had made views of its flat
    buffers: the next update must continue from the LOADED state (a resumed run), not from stale moments."""
    import copy
    from brl_amd.models import make_forward_pass
    from brl_amd.update import make_update_step
    from tests.test_update_cpu import CFG, fake_batch
    fp = make_forward_pass("relu", "DeepMind")
    cfg = dict(CFG, minibatch_size=256, update_epochs=1)
    upd = make_update_step(cfg, fp)
    data = []
    for seed in (31, 32):
        tb, adv, tgt = fake_batch(4, 256, seed=seed)
        data.append((type(tb)(*[x.cuda() for x in tb]), adv.cuda(), tgt.cuda()))
    net = fp.init(3, device="cuda")
    rs, _ = upd((net, None, None, None, 0, 5), *data[0])
    snap_w = copy.deepcopy(net.state_dict())
    snap_o = copy.deepcopy(rs[1]["opt"].state_dict())
    rs, _ = upd(rs, *data[1])
    want = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    rs, _ = upd(rs, *data[0])                       # move on, then rewind weights AND optimizer and redo the second update
    net.load_state_dict(snap_w)
    rs[1]["opt"].load_state_dict(snap_o)
    rs = (net, rs[1], None, None, 0, 6)
    rs, _ = upd(rs, *data[1])
    got = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(got, want, atol=1e-7, rtol=1e-6), float((got - want).abs().max())
    assert {int(st["step"]) for st in rs[1]["opt"].state.values()} == {8}


def test_fused_update_helpers_match_torch():
    """brl_act_bwd_colsum + brl_bias_finalize_ex, brl_adam_clip_fin_gather and brl_mb_gather_bind / _dev against their torch counterparts."""
    import ctypes as C
    from brl_amd import _capi
    L, dev = _capi.lib(), torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(1)
    # activation backward + bias gradient, ragged row counts
    for rows, cols, act in ((1024, 1024, 0), (300, 40, 1), (64, 100, 0), (250, 1024, 1), (7, 8, 0)):
        dh = torch.randn(rows, cols, device=dev, generator=g)
        h = torch.randn(rows, cols, device=dev, generator=g).clamp_(-0.99, 0.99)
        want_dz = dh * (h > 0) if act == 0 else dh * (1 - h * h)
        db = torch.empty(cols, device=dev)
        tiles = (rows + 15) // 16
        scratch = torch.empty(tiles * cols, device=dev)
        got = dh.clone()
        _capi.check(L.brl_act_bwd_colsum(0, got.data_ptr(), h.data_ptr(), rows, cols, cols, act, scratch.data_ptr(), s))
        _capi.check(L.brl_bias_finalize_ex(0, 1, (C.c_void_p * 1)(scratch.data_ptr()), (C.c_int64 * 1)(cols), (C.c_int64 * 1)(tiles),
                                           (C.c_void_p * 1)(db.data_ptr()), s))
        assert torch.allclose(got, want_dz, rtol=1e-6, atol=0) and torch.allclose(db, want_dz.sum(0), atol=1e-3, rtol=1e-5)
        assert act == 1 or torch.equal(got, want_dz)
    # clip + Adam (single rank: the sums of partials ride in the norm launch), three steps, against torch.optim.Adam + clip_grad_norm_
    n, tail, tiles = 4096 + 8, 8, 3                      # the last 8 gradients come as 3 tiles of partial sums
    p0 = torch.randn(n, device=dev, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, eps=1e-5)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step, scratch, idx, norm = torch.zeros((), device=dev), torch.empty(2048, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(1, device=dev)
    lr_dev = torch.full((1,), 1e-3, device=dev)   # the third step reads the learning rate from device memory
    for it in range(3):
        grad = torch.randn(n, device=dev, generator=g) * (0.001 if it == 1 else 1.0)   # one step below the clip threshold
        parts = torch.randn(tiles, tail, device=dev, generator=g) * (0.001 if it == 1 else 1.0)
        full = grad.clone()
        full[n - tail:] = parts.sum(0)
        ref.grad = full.clone()
        want_norm = torch.nn.utils.clip_grad_norm_([ref], 0.5)
        opt.step()
        grad[n - tail:] = float("nan")                   # (written by the norm launch's finalize blocks)
        _capi.check(L.brl_adam_clip_fin_gather(0, p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, step.data_ptr(), 1e-3,
                                               lr_dev.data_ptr() if it == 2 else None, 0.9, 0.999, 1e-5, 0.5, scratch.data_ptr(), scratch.numel(),
                                               idx.data_ptr(), norm.data_ptr(), None, 0, 1, (C.c_void_p * 1)(parts.data_ptr()),
                                               (C.c_int64 * 1)(tail), (C.c_int64 * 1)(tiles), (C.c_void_p * 1)(grad[n - tail:].data_ptr()), s))
        assert torch.allclose(grad[n - tail:], full[n - tail:], atol=1e-6)
        assert torch.allclose(norm[0], want_norm, rtol=1e-5)
        assert torch.allclose(p, ref.detach(), atol=2e-6), float((p - ref.detach()).abs().max())
    assert int(idx.item()) == 3 and float(step.item()) == 3.0
    # minibatch gather: arguments bound in device memory (what the captured minibatch step launches)
    from tests.test_update_cpu import fake_batch
    tb, adv, tgt = fake_batch(4, 64, seed=5)
    flat = type(tb)(*[x.reshape((256,) + x.shape[2:]).cuda() for x in tb])
    adv, tgt = adv.reshape(-1).cuda(), tgt.reshape(-1).cuda()
    perm = torch.randperm(256, device=dev, generator=g)
    tp = _capi.TransitionPtrs()
    for name in _capi.TransitionPtrs._names:
        t = getattr(flat, name)
        setattr(tp, name, (t.view(torch.uint8) if t.dtype == torch.bool else t).data_ptr())
    B = 64
    x0, mask, act = torch.empty(B, 480, device=dev), torch.empty(B, 38, dtype=torch.uint8, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
    ov, olp, ga, tg = (torch.empty(B, device=dev) for _ in range(4))
    mbi = torch.full((1,), 2, dtype=torch.int32, device=dev)
    gargs = torch.zeros(256, dtype=torch.uint8, device=dev)
    _capi.check(L.brl_mb_gather_bind(0, C.byref(tp), adv.data_ptr(), tgt.data_ptr(), perm.data_ptr(), mbi.data_ptr(), B, x0.data_ptr(),
                                     mask.data_ptr(), act.data_ptr(), ov.data_ptr(), olp.data_ptr(), ga.data_ptr(), tg.data_ptr(),
                                     256 // B, gargs.data_ptr(), s))
    mbi.fill_(1)   # read by the launch, not by the bind
    _capi.check(L.brl_mb_gather_dev(0, gargs.data_ptr(), B, s))
    rows = perm[B:2 * B]
    assert torch.equal(x0, flat.obs[rows].float()) and torch.equal(mask.bool(), flat.legal_action_mask[rows])
    assert torch.equal(act, flat.action[rows]) and torch.equal(ov, flat.value[rows]) and torch.equal(olp, flat.log_prob[rows])
    assert torch.equal(ga, adv[rows]) and torch.equal(tg, tgt[rows])
    # ... and as extra workgroups of the Adam launches (both the single-rank and the shard form): the norm launch advances the
    # counter, the sweep gathers THAT minibatch beside the parameter update; a minibatch past the bound permutation (4 here) is skipped
    geom = _capi.ShardGeom()
    geom.nbuckets, geom.world, geom.nsub = 1, 2, 4
    geom.off[0], geom.len[0] = 0, n // 2
    part = torch.empty(8, device=dev)
    p2, m2, v2 = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step2 = torch.zeros((), device=dev)
    for want_idx in (2, 3, 4):
        grad = torch.randn(n, device=dev, generator=g)
        if want_idx == 3:
            _capi.check(L.brl_adam_shard_norm(0, grad.data_ptr(), C.byref(geom), 0, 2, 1.0, part.data_ptr(), step2.data_ptr(), mbi.data_ptr(), s))
            _capi.check(L.brl_adam_shard_apply(0, p2.data_ptr(), grad.data_ptr(), m2.data_ptr(), v2.data_ptr(), C.byref(geom), 0, 2, part.data_ptr(),
                                               step2.data_ptr(), 1e-3, None, 0.9, 0.999, 1e-5, 0.5, 1.0, norm.data_ptr(), gargs.data_ptr(), B, s))
        else:
            parts = torch.zeros(1, tail, device=dev)
            _capi.check(L.brl_adam_clip_fin_gather(0, p2.data_ptr(), grad.data_ptr(), m2.data_ptr(), v2.data_ptr(), n, step2.data_ptr(), 1e-3, None,
                                                   0.9, 0.999, 1e-5, 0.5, scratch.data_ptr(), scratch.numel(), mbi.data_ptr(), norm.data_ptr(),
                                                   gargs.data_ptr(), B, 1, (C.c_void_p * 1)(parts.data_ptr()), (C.c_int64 * 1)(tail),
                                                   (C.c_int64 * 1)(1), (C.c_void_p * 1)(grad[n - tail:].data_ptr()), s))
        assert int(mbi.item()) == want_idx
        rows = perm[min(want_idx, 3) * B:(min(want_idx, 3) + 1) * B]     # (index 4: nothing gathered, minibatch 3 stays)
        assert torch.equal(x0, flat.obs[rows].float()) and torch.equal(act, flat.action[rows]) and torch.equal(tg, tgt[rows])
    assert float(step2.item()) == 3.0 and not torch.equal(p2, p0)


@pytest.mark.parametrize("B,H,act", [(1024, 1024, 0), (1000, 1024, 1), (48, 256, 0), (17, 512, 1)])
def test_head_kernels_match_torch_and_the_separate_launches(B, H, act):
    """brl_ppo_heads_loss_split / brl_ppo_heads_bwd / brl_ppo_stats_gram / brl_bias_finalize_ex (the 39-column head of one PPO
    minibatch step) against float64 torch and against the launches they replace: heads = h W^T + b; d(heads), statistics partials
    and the illegal-action Gram matrix equal to brl_ppo_loss / brl_ppo_stats on the SAME heads (bit-identical d(heads)); dW_h, db_h,
    dh * act'(h) and its column sums vs torch; the illegal-action norm vs an SVD."""
    from brl_amd import _capi
    L, dev = _capi.lib(), torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(B + H)
    rn = lambda *shape: torch.randn(*shape, device=dev, generator=g)  # noqa: E731
    h = rn(B, H).relu_() if act == 0 else rn(B, H).tanh_()
    Wh, bh = rn(39, H) / H ** 0.5, rn(39) * 0.1
    mask = (torch.rand(B, 38, device=dev, generator=g) < 0.6).to(torch.uint8)
    mask[:, 0] = 1
    action = torch.multinomial(mask.float(), 1, generator=g)[:, 0].to(torch.int32)
    old_v, old_lp, gae, tgt = rn(B) * 0.3, -rn(B).abs() - 0.1, rn(B), rn(B) * 0.3
    groups, lgroups = (B + 15) // 16, (B + 3) // 4     # 16-row tiles of the column sums; 4-sample groups of the loss launch
    heads, dheads = torch.empty(B, 39, device=dev), torch.empty(B, 39, device=dev)
    partials, gram_p = torch.empty(lgroups, 8, device=dev), torch.empty(lgroups, 1444, device=dev)
    def loss_of(hd, adv_):
        """brl_ppo_loss on a given [B,39] heads matrix (logits = columns 0..37, row stride 39; value = column 38) -> d(heads) [B,39]"""
        dl, dv = torch.empty(B, 38, device=dev), torch.empty(B, device=dev)
        p2_, illp_ = torch.empty((B + 3) // 4, 8, device=dev), torch.empty(B, 38, device=dev)
        val = hd[:, 38].contiguous()
        _capi.check(L.brl_ppo_loss(0, hd.data_ptr(), 39, val.data_ptr(), mask.data_ptr(), action.data_ptr(), old_v.data_ptr(), old_lp.data_ptr(),
                                   adv_.contiguous().data_ptr(), tgt.data_ptr(), B, 0.2, 0.5, 0.001, 1, 1, dl.data_ptr(), dv.data_ptr(),
                                   p2_.data_ptr(), illp_.data_ptr(), s))
        return torch.cat([dl, dv[:, None]], 1), p2_, illp_

    for rscale in (0, 1):
        want = (h.double() @ Wh.double().t() + bh.double())
        adv = (gae - gae.mean()) / (gae.std(unbiased=False) + 1e-8) if rscale else gae
        # the heads product split over K across workgroups, then the loss on bias + parts: the same heads up to the order of the
        # fp32 sums, and d(heads) / partials / Gram partials equal to the separate loss launch on ITS heads
        first = None
        for ksplit in (1, 4) if H % 64 == 0 else (1, 2):
            heads3, dheads3 = torch.empty(B, 39, device=dev), torch.empty(B, 39, device=dev)
            partials3, gram_p3 = torch.empty(lgroups, 8, device=dev), torch.empty(lgroups, 1444, device=dev)
            hparts = torch.full((ksplit, B, 39), float("nan"), device=dev)
            _capi.check(L.brl_ppo_heads_loss_split(0, h.data_ptr(), H, Wh.data_ptr(), bh.data_ptr(), H, mask.data_ptr(), action.data_ptr(),
                                                   old_v.data_ptr(), old_lp.data_ptr(), gae.data_ptr(), tgt.data_ptr(), B, 0.2, 0.5, 0.001,
                                                   1, 1, rscale, heads3.data_ptr(), dheads3.data_ptr(), partials3.data_ptr(),
                                                   gram_p3.data_ptr(), hparts.data_ptr(), ksplit, s))
            assert not torch.isnan(hparts).any()
            assert float((heads3.double() - want).abs().max()) < 2e-5          # fp32 sum of 1024 products
            dh3, p2, illp = loss_of(heads3, adv)
            if rscale == 0:
                assert torch.equal(dheads3, dh3)
            else:   # the in-kernel mean / std differ from torch's in the last bits
                assert torch.allclose(dheads3, dh3, rtol=1e-4, atol=1e-9)
            if first is None:
                first = (dheads3, partials3, gram_p3)
                heads.copy_(heads3); dheads.copy_(dheads3); partials.copy_(partials3); gram_p.copy_(gram_p3)
            assert torch.allclose(dheads3, first[0], rtol=1e-3, atol=1e-6)
            assert torch.allclose(partials3.sum(0), first[1].sum(0), rtol=1e-4, atol=1e-5)
            assert torch.allclose(gram_p3.sum(0), first[2].sum(0), rtol=1e-4, atol=1e-7)
        _, p2, illp = loss_of(heads, adv)
        assert torch.allclose(partials.sum(0), p2.sum(0), rtol=1e-4, atol=1e-5)
        gram = illp.double().t() @ illp.double()
        assert torch.allclose(gram_p.sum(0).double().reshape(38, 38), gram, rtol=1e-4, atol=1e-7)
        out_new, out_old, vec = torch.zeros(8, device=dev), torch.zeros(8, device=dev), torch.zeros(40, device=dev)
        _capi.check(L.brl_ppo_stats_gram(0, partials.data_ptr(), lgroups, B, gram_p.data_ptr(), lgroups, 0.5, 0.001, 0.0, out_new.data_ptr(),
                                         None, vec.data_ptr(), s))
        gram32 = gram.float().contiguous()
        _capi.check(L.brl_ppo_stats(0, p2.data_ptr(), B, gram32.data_ptr(), 0.5, 0.001, out_old.data_ptr(), s))
        assert torch.allclose(out_new, out_old, rtol=2e-4, atol=1e-6), (out_new, out_old)
        sv = torch.linalg.svdvals(illp.double())[0]
        assert abs(float(out_new[6]) - float(sv) / 2) < 1e-4 * float(sv) + 1e-7     # jnp.linalg.norm(P, ord=2) / 2
        u, sg, vt = torch.linalg.svd(illp.double(), full_matrices=False)
        v1 = vt[0] * torch.sign(vt[0].sum())
        assert abs(float(vec[38]) - float(sg[0])) < 1e-4 * float(sg[0]) + 1e-7
        assert float((vec[:38].double() - v1).abs().max()) < 2e-3                      # (l2 / l1)^256 away from the Perron vector
    # gradient of the illegal-action norm (src/update.py:146-152) from the power iteration's v1 / sigma_1 vs autograd through an SVD
    lg = heads[:, :38].detach().clone().requires_grad_(True)
    (0.7 * torch.linalg.matrix_norm(torch.softmax(lg, -1) * (mask == 0), ord=2) / 2).backward()
    add = torch.zeros(B, 39, device=dev)
    _capi.check(L.brl_ppo_illegal_grad(0, heads.data_ptr(), mask.data_ptr(), vec.data_ptr(), 0.7, B, add.data_ptr(), s))
    assert float(add[:, 38].abs().max()) == 0 and float(lg.grad.abs().max()) > 0
    assert float((add[:, :38] - lg.grad).abs().max()) < 2e-3 * float(lg.grad.abs().max())
    # backward of the head
    nsplit = (B + 63) // 64
    dwp, dbp = torch.empty(nsplit, 39 * H, device=dev), torch.empty(nsplit, 39, device=dev)
    dh, ts = torch.empty(B, H, device=dev), torch.empty(groups, H, device=dev)
    # (with the step's statistics sums accumulated by the same launches, into row 2 of per-update buffers)
    row = torch.full((1,), 2, dtype=torch.int32, device=dev)
    ssum, gsum, rows_out = torch.zeros(4, 8, device=dev), torch.zeros(4, 1444, device=dev), torch.zeros(4, 8, device=dev)
    _capi.check(L.brl_ppo_heads_bwd(0, dheads.data_ptr(), h.data_ptr(), H, Wh.data_ptr(), B, H, act, nsplit, dwp.data_ptr(),
                                    dbp.data_ptr(), dh.data_ptr(), ts.data_ptr(), partials.data_ptr(), gram_p.data_ptr(), lgroups,
                                    row.data_ptr(), ssum.data_ptr(), gsum.data_ptr(), s))
    _capi.check(L.brl_ppo_stats_rows(0, ssum.data_ptr(), gsum.data_ptr(), 4, B, 0.5, 0.001, 0.0, rows_out.data_ptr(), s))
    assert torch.allclose(rows_out[2], out_new, rtol=1e-5, atol=1e-7), (rows_out[2], out_new)   # the per-step launch's row
    assert float(ssum[[0, 1, 3]].abs().max()) == 0 and float(gsum[[0, 1, 3]].abs().max()) == 0
    import ctypes as C
    gW, gb, gbias = torch.empty(39, H, device=dev), torch.empty(39, device=dev), torch.empty(H, device=dev)
    parts = (C.c_void_p * 3)(dwp.data_ptr(), dbp.data_ptr(), ts.data_ptr())
    cols, tiles = (C.c_int64 * 3)(39 * H, 39, H), (C.c_int64 * 3)(nsplit, nsplit, groups)
    outs = (C.c_void_p * 3)(gW.data_ptr(), gb.data_ptr(), gbias.data_ptr())
    _capi.check(L.brl_bias_finalize_ex(0, 3, parts, cols, tiles, outs, s))
    d64, h64 = dheads.double(), h.double()
    scale = float(d64.abs().max())
    assert float((gW.double() - d64.t() @ h64).abs().max()) < 1e-4 * scale * 32
    assert float((gb.double() - d64.sum(0)).abs().max()) < 1e-4 * scale * 32
    deriv = (h64 > 0).double() if act == 0 else 1 - h64 * h64
    want_dh = (d64 @ Wh.double()) * deriv
    assert float((dh.double() - want_dh).abs().max()) < 1e-5 * scale * 8
    assert float((gbias.double() - dh.double().sum(0)).abs().max()) < 1e-4 * scale * 32
    # the activation-derivative pass of the layers below, both activations
    dz = rn(B, H)
    want = dz * ((h > 0).float() if act == 0 else 1 - h * h)
    got = dz.clone()
    _capi.check(L.brl_act_bwd_colsum(0, got.data_ptr(), h.data_ptr(), B, H, H, act, ts.data_ptr(), s))
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert torch.allclose(ts.sum(0), want.sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("layout,epi,M,N,K,act", [
    (0, 1, 1024, 1024, 1024, 0), (0, 1, 1024, 1024, 480, 0), (0, 1, 200, 72, 64, 1), (0, 0, 48, 256, 480, 0),
    (1, 2, 1024, 1024, 1024, 0), (1, 2, 100, 36, 96, 1), (1, 2, 1000, 256, 256, 0), (1, 0, 64, 64, 32, 0),
    (2, 3, 1024, 1024, 1024, 0), (2, 3, 68, 132, 1000, 0), (2, 0, 1024, 480, 1024, 0), (2, 3, 4, 4, 4, 0)])
def test_mlp_gemm_matches_float64(layout, epi, M, N, K, act):
    """brl_mlp_gemm (csrc/mlp_gemm.hpp: the step's fp32 MFMA products with fused epilogues) against a float64 torch product on the
    same operands: every layout (NT forward, NN dh, TN dW), every epilogue (bias + ReLU / tanh; activation derivative + 64-row
    column sums; tile square sums), the step's shapes, edge tiles and a K tail.  fp32 k-ordered fma chains: 2e-4 * max|ref| at
    K = 1024 (the same class of result as the library GEMM it replaces, src/update.py:86-178)."""
    from brl_amd import _capi
    L = _capi.lib()
    g = torch.Generator(device="cuda").manual_seed(layout * 1000 + M + N + K)
    r = lambda *sh: (torch.rand(sh, device="cuda", generator=g) * 2 - 1)  # noqa: E731
    akc, bkc = layout != 2, layout == 0
    A = r(M, K) if akc else r(K, M)
    Bm = r(N, K) if bkc else r(K, N)
    bias, gate = r(N), r(M, N)
    C = torch.full((M, N), float("nan"), device="cuda")
    tm, tn = (M + 63) // 64, (N + 63) // 64
    colsum = torch.full((tm, N), float("nan"), device="cuda")
    sqsum = torch.zeros((tm * ((N + 31) // 32),), device="cuda")   # one word per tile: sized for 64 x 32 tiles (the library picks the width)
    s = torch.cuda.current_stream().cuda_stream
    _capi.check(L.brl_mlp_gemm(0, layout, epi, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), C.data_ptr(), N, M, N, K, act,
                               bias.data_ptr(), gate.data_ptr(), N, colsum.data_ptr(), sqsum.data_ptr(), s))
    torch.cuda.synchronize()
    a64 = A.double() if akc else A.double().t()
    b64 = Bm.double().t() if bkc else Bm.double()
    ref = a64 @ b64
    if epi == 1:
        ref = ref + bias.double()
        ref = ref.clamp_min(0) if act == 0 else ref.tanh()
    if epi == 2:
        ref = ref * ((gate > 0).double() if act == 0 else (1 - gate.double() ** 2))
    tol = 2e-4 * max(1.0, float(ref.abs().max())) * (K / 1024 + 1) ** 0.5
    assert float((C.double() - ref).abs().max()) < tol
    if epi == 2:   # column sums of what was STORED, per 64-row tile, in a fixed order: compare with the fp64 sums of C itself
        want = torch.stack([C[64 * t:64 * t + 64].double().sum(0) for t in range(tm)])
        assert float((colsum.double() - want).abs().max()) < 1e-3
    if epi == 3:
        assert abs(float(sqsum.double().sum()) - float((C.double() ** 2).sum())) <= 1e-5 * max(1.0, float((C.double() ** 2).sum()))
    # a second launch gives the same bits (no atomics, fixed summation order)
    C2 = torch.empty_like(C)
    _capi.check(L.brl_mlp_gemm(0, layout, epi, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), C2.data_ptr(), N, M, N, K, act,
                               bias.data_ptr(), gate.data_ptr(), N, colsum.data_ptr(), sqsum.data_ptr(), s))
    assert torch.equal(C, C2)


@pytest.mark.parametrize("layout,epi,M,N,K,act,ws", [
    (0, 1, 8192, 1024, 1024, 0, False), (0, 1, 4096, 1024, 480, 0, False), (0, 1, 1024, 1024, 1024, 0, True), (0, 1, 1024, 1024, 1024, 0, False),
    (0, 1, 200, 72, 64, 1, True), (0, 0, 48, 256, 992, 0, True), (0, 0, 64, 64, 32, 0, False),
    (1, 2, 1024, 1024, 1024, 0, True), (1, 2, 100, 36, 96, 1, True), (1, 2, 1000, 256, 256, 0, False),
    (2, 0, 1024, 1024, 1024, 0, True), (2, 0, 68, 132, 992, 0, True), (2, 0, 1024, 480, 1024, 0, False)])
def test_mlp_gemm_x3_beats_the_exact_kernels_error(layout, epi, M, N, K, act, ws):
    """brl_mlp_gemm_x3 (csrc/mlp_gemm_x3.hpp: the fp32 product as six bf16 MFMA products of three exact bf16 pieces per operand,
    128 x 128 tiles, optionally K divided among workgroups) on the SAME operands as brl_mlp_gemm, both against a float64 product: every
    layout and epilogue it offers, the rollout's and the step's shapes, edge tiles, one chunk and odd chunk counts (k is a multiple of 32:
    other k are refused — brl_mlp_gemm takes them), with and without the split-K workspace.
    The bound is the exact kernel's own (2e-4 * max|ref| at K = 1024) AND its error on these very inputs: max |err| of bf16x3 <= the exact
    kernel's (measured 0.07 - 0.64 x: scripts/micro/gemm_x3_test.hip) + one ulp of slack for the tiny shapes.  Deterministic: the same bits twice."""
    import ctypes as C
    from brl_amd import _capi
    L = _capi.lib()
    g = torch.Generator(device="cuda").manual_seed(layout * 1000 + M + N + K)
    r = lambda *sh: (torch.rand(sh, device="cuda", generator=g) * 2 - 1)  # noqa: E731
    akc, bkc = layout != 2, layout == 0
    A = r(M, K) if akc else r(K, M)
    Bm = r(N, K) if bkc else r(K, N)
    bias, gate = r(N), r(M, N)
    tm = (M + 63) // 64
    s = torch.cuda.current_stream().cuda_stream
    need = C.c_int64(0)
    _capi.check(L.brl_mlp_gemm_x3_workspace(M, N, K, C.byref(need)))
    assert need.value >= 0 and (need.value > 0) == (((M + 127) // 128) * ((N + 127) // 128) < 256 and K >= 128)
    work = torch.zeros(max(need.value, 256) // 4, dtype=torch.int32, device="cuda") if ws else None

    def run(fn_x3):
        Cm = torch.full((M, N), float("nan"), device="cuda")
        colsum = torch.full((tm, N), float("nan"), device="cuda")
        if fn_x3:
            _capi.check(L.brl_mlp_gemm_x3(0, layout, epi, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), Cm.data_ptr(), N, M, N, K, act,
                                          bias.data_ptr(), gate.data_ptr(), N, colsum.data_ptr(), work.data_ptr() if ws else None,
                                          need.value if ws else 0, s))
        else:
            _capi.check(L.brl_mlp_gemm(0, layout, epi, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), Cm.data_ptr(), N, M, N, K, act,
                                       bias.data_ptr(), gate.data_ptr(), N, colsum.data_ptr(), None, s))
        torch.cuda.synchronize()
        return Cm, colsum
    C3, cs3 = run(True)
    C1, _ = run(False)
    a64 = A.double() if akc else A.double().t()
    b64 = Bm.double().t() if bkc else Bm.double()
    ref = a64 @ b64
    if epi == 1:
        ref = ref + bias.double()
        ref = ref.clamp_min(0) if act == 0 else ref.tanh()
    if epi == 2:
        ref = ref * ((gate > 0).double() if act == 0 else (1 - gate.double() ** 2))
    e3, e1 = float((C3.double() - ref).abs().max()), float((C1.double() - ref).abs().max())
    assert e3 < 2e-4 * max(1.0, float(ref.abs().max())) * (K / 1024 + 1) ** 0.5
    assert e3 <= e1 + 2.0 ** -22 * max(1.0, float(ref.abs().max())), (e3, e1)
    if epi == 2:   # column sums of what was STORED, per 64-row tile
        want = torch.stack([C3[64 * t:64 * t + 64].double().sum(0) for t in range(tm)])
        assert float((cs3.double() - want).abs().max()) < 1e-3
    C3b, _ = run(True)
    assert torch.equal(C3, C3b)
    if ws:   # every ticket is back at zero: the workspace is ready for the next product
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        assert int(work[:tiles].abs().sum()) == 0
    # refusals: k not a multiple of 32, an epilogue it does not offer
    assert L.brl_mlp_gemm_x3(0, layout, 0, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), C3.data_ptr(), N, M, N, K - 4, act, None, None, 0,
                             None, None, 0, s) == -1
    assert L.brl_mlp_gemm_x3(0, 2, 3, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), C3.data_ptr(), N, M, N, K, act, None, None, 0,
                             None, None, 0, s) == _capi.BRL_E_ARG if hasattr(_capi, "BRL_E_ARG") else True


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_mlp_gemm_group_matches_float64(layout):
    """brl_mlp_gemm_group: several products of one layout in ONE launch (the FAIR step's twelve weight gradients with layout TN, the
    heads' 39 rows out of a [K, 40] array among them) — every output against float64 with brl_mlp_gemm's bound (the same tile code),
    shapes from one 4 x 4 x 4 product to 200 x 680 x 1024; more than 16 products are refused."""
    import ctypes as C
    from brl_amd import _capi
    L = _capi.lib()
    g = torch.Generator(device="cuda").manual_seed(77 + layout)
    r = lambda *sh: (torch.rand(sh, device="cuda", generator=g) * 2 - 1)  # noqa: E731
    akc, bkc = layout != 2, layout == 0
    shapes = [(200, 200, 1024)] * 9 + [(200, 680, 1024), (200, 480, 1024), (64, 36, 52), (4, 4, 4)]
    As = [r(M, K) if akc else r(K, M) for M, N, K in shapes]
    Bs = [r(N, K) if bkc else r(K, N) for M, N, K in shapes]
    if layout == 2:     # the FAIR heads' weight gradient: m = 39 rows out of a [K, 40] array (lda = m rounded up to 4)
        shapes = shapes + [(39, 200, 1024)]
        As.append(r(1024, 40)[:, :39])
        Bs.append(r(1024, 200))
    Cs = [torch.full((M, N), float("nan"), device="cuda") for M, N, K in shapes]
    n = len(shapes)
    vp, i64 = C.c_void_p * n, C.c_int64 * n
    s = torch.cuda.current_stream().cuda_stream
    _capi.check(L.brl_mlp_gemm_group(0, layout, n, vp(*[t.data_ptr() for t in As]), i64(*[t.stride(0) for t in As]),
                                     vp(*[t.data_ptr() for t in Bs]), i64(*[t.stride(0) for t in Bs]), vp(*[t.data_ptr() for t in Cs]),
                                     i64(*[sh[1] for sh in shapes]), i64(*[sh[0] for sh in shapes]), i64(*[sh[1] for sh in shapes]),
                                     i64(*[sh[2] for sh in shapes]), s))
    for A, Bm, Cg, (M, N, K) in zip(As, Bs, Cs, shapes):
        ref = (A.double() if akc else A.double().t()) @ (Bm.double().t() if bkc else Bm.double())
        assert float((Cg.double() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())) * (K / 1024 + 1) ** 0.5
    with pytest.raises(_capi.BrlError):
        _capi.check(L.brl_mlp_gemm_group(0, layout, 17, None, None, None, None, None, None, None, None, None, s))


@pytest.mark.parametrize("M,N,K,npx", [(8192, 1024, 1024, 3), (8192, 1024, 480, 1), (5000, 256, 480, 1), (4097, 128, 32, 3), (300, 384, 96, 3)])
def test_linear_x3p_matches_float64_and_beats_the_exact_kernel(M, N, K, npx):
    """brl_split_planes + brl_linear_x3p (csrc/mlp_linear_x3p.hpp: the large-batch inference layer on operands pre-split into bf16 planes;
    npx = 1: a 0/1 input as ONE bf16 plane, the observation): the planes reproduce their fp32 source exactly (hi + mid + lo == x); the
    layer's fp32 output against float64 within the exact kernel's bound AND no worse than brl_mlp_gemm (exact fp32 MFMA chain) on the same
    operands; the output planes == the fp32 output, bit for bit; edge row tiles; the same bits twice; the oracle shim's restatement
    on a few rows; refusals."""
    import ctypes as C
    from brl_amd import _capi
    from oracle.binding import shim_path
    import oracle
    L = _capi.lib()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    s = torch.cuda.current_stream().cuda_stream
    x = (torch.rand((M, K), device="cuda", generator=g) < 0.1).float() if npx == 1 else torch.rand((M, K), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((N, K), device="cuda", generator=g) * 2 - 1) * 0.05
    b = torch.rand(N, device="cuda", generator=g) - 0.5
    wp = torch.empty((3, N, K), dtype=torch.bfloat16, device="cuda")
    _capi.check(L.brl_split_planes(0, w.data_ptr(), N * K, wp.data_ptr(), N * K, s))
    assert torch.equal((wp[0].float() + wp[1].float()) + wp[2].float(), w)
    if npx == 3:
        xp = torch.empty((3, M, K), dtype=torch.bfloat16, device="cuda")
        _capi.check(L.brl_split_planes(0, x.data_ptr(), M * K, xp.data_ptr(), M * K, s))
        assert torch.equal((xp[0].float() + xp[1].float()) + xp[2].float(), x)
    else:
        xp = x.to(torch.bfloat16)

    def run(with_y, with_planes):
        y = torch.full((M, N), float("nan"), device="cuda") if with_y else None
        yp = torch.zeros((3, M, N), dtype=torch.bfloat16, device="cuda") if with_planes else None
        _capi.check(L.brl_linear_x3p(0, xp.data_ptr(), npx, K, M * K if npx == 3 else 0, wp.data_ptr(), K, N * K, b.data_ptr(), 1,
                                     y.data_ptr() if with_y else None, N, yp.data_ptr() if with_planes else None, N, M * N, M, N, K, s))
        torch.cuda.synchronize()
        return y, yp
    y, yp = run(True, True)
    ref = (x.double() @ w.double().t() + b.double()).clamp_min(0)
    e3 = float((y.double() - ref).abs().max())
    y1 = torch.empty((M, N), device="cuda")
    _capi.check(L.brl_mlp_gemm(0, 0, 1, x.data_ptr(), K, w.data_ptr(), K, y1.data_ptr(), N, M, N, K, 0, b.data_ptr(), None, 0, None, None, s))
    e1 = float((y1.double() - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert e3 < 2e-4 * scale * (K / 1024 + 1) ** 0.5 and e3 <= e1 + 2.0 ** -22 * scale, (e3, e1)
    assert torch.equal((yp[0].float() + yp[1].float()) + yp[2].float(), y)
    y_only, _ = run(True, False)
    _, p_only = run(False, True)
    assert torch.equal(y_only, y) and torch.equal(p_only, yp)
    # the oracle shim's restatement (float64 over the planes' sums) on the first rows
    oracle.build()
    shim = C.CDLL(shim_path())
    m = 3
    xs, ws, bs = xp.cpu().contiguous().view(torch.int16).numpy(), wp.cpu().contiguous().view(torch.int16).numpy(), b.cpu().numpy()
    ys = np.zeros((m, N), np.float32)
    shim.brl_linear_x3p.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    assert shim.brl_linear_x3p(0, xs.ctypes.data, npx, K, M * K if npx == 3 else 0, ws.ctypes.data, K, N * K, bs.ctypes.data, 1, ys.ctypes.data, N,
                               None, 0, 0, m, N, K, None) == 0
    assert np.abs(ys - y[:m].cpu().numpy()).max() < 2e-4 * scale
    # strided operands and outputs: rows of x / W / y inside wider arrays, planes further apart than they are long
    if M <= 5000:
        ldx, ldw, ldy = K + 8, K + 16, N + 4
        xs2 = torch.zeros((npx, M, ldx), dtype=torch.bfloat16, device="cuda")
        xs2[:, :, :K] = xp if npx == 3 else xp[None]
        ws2 = torch.zeros((3, N + 5, ldw), dtype=torch.bfloat16, device="cuda")
        ws2[:, :N, :K] = wp
        y2 = torch.full((M, ldy), 7.0, device="cuda")
        yp2 = torch.zeros((3, M + 1, N + 8), dtype=torch.bfloat16, device="cuda")
        _capi.check(L.brl_linear_x3p(0, xs2.data_ptr(), npx, ldx, M * ldx, ws2.data_ptr(), ldw, (N + 5) * ldw, b.data_ptr(), 1, y2.data_ptr(), ldy,
                                     yp2.data_ptr(), N + 8, (M + 1) * (N + 8), M, N, K, s))
        torch.cuda.synchronize()
        assert torch.equal(y2[:, :N], y) and bool((y2[:, N:] == 7.0).all()) and torch.equal(yp2[:, :M, :N], yp)
        assert not bool(yp2[:, M:].any()) and not bool(yp2[:, :, N:].any())
    # refusals: n not a multiple of 128, k not of 32, no output at all
    for bad in ((M, N - 64, K), (M, N, K - 8)):
        assert L.brl_linear_x3p(0, xp.data_ptr(), npx, K, M * K, wp.data_ptr(), K, N * K, b.data_ptr(), 1, y.data_ptr(), N, None, 0, 0, *bad, s) == -1
    assert L.brl_linear_x3p(0, xp.data_ptr(), npx, K, M * K, wp.data_ptr(), K, N * K, b.data_ptr(), 1, None, 0, None, 0, 0, M, N, K, s) == -1


@pytest.mark.parametrize("model", ["DeepMind", "DeepMind_6"])
def test_inference_snapshot_large_batch_paths_against_float64(model, monkeypatch):
    """The three ways an fp32 forward of 8192 observations runs its hidden layers — brl_linear_x3p (the default: bf16x3 products on operands
    pre-split into planes, the 0/1 observation as one bf16 plane), brl_mlp_gemm_x3 (BRL_INFERENCE_PLANES=0: the split in registers) and the
    library's exact fp32 GEMM (inference_gemm = "library") — through the WHOLE network (src/models.py:23-33) against the module in float64:
    the two bf16x3 paths are no further from float64 than the library path (fp32-grade end to end), and follow refreshed weights."""
    from brl_amd.models import InferenceSnapshot, make_forward_pass
    fp = make_forward_pass("relu", model)
    net = fp.init(11, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    obs = torch.rand((8192, 480), device="cuda", generator=g) < 0.12
    with torch.no_grad():
        ref = fp.init(11, device="cpu").double()
        ref.load_state_dict({k: t.double().cpu() for k, t in net.state_dict().items()})
        lg64, v64 = ref(obs.double().cpu())
    want = torch.cat([lg64, v64[:, None]], 1)

    def run(gemm, planes):
        monkeypatch.setenv("BRL_INFERENCE_PLANES", "1" if planes else "0")
        snap = InferenceSnapshot.make(net, gemm=gemm)
        assert (snap.wp is not None) == (planes and gemm != "library")
        with torch.no_grad():
            out = snap.heads(obs)
        return snap, float((out.double().cpu() - want).abs().max())
    snap_p, e_planes = run("bf16x3", True)
    _, e_x3 = run("bf16x3", False)
    _, e_lib = run("library", False)
    scale = max(1.0, float(want.abs().max()))
    assert e_lib < 2e-4 * scale and e_planes <= e_lib * 1.05 + 1e-7 * scale and e_x3 <= e_lib * 1.05 + 1e-7 * scale, (e_planes, e_x3, e_lib)
    # new weights: refresh re-splits the planes INTO the same tensors (their addresses sit in captured graphs)
    ptrs = [w.data_ptr() for w in snap_p.wp]
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.01)
        snap_p.refresh(net)
        ref.load_state_dict({k: t.double().cpu() for k, t in net.state_dict().items()})
        lg64, v64 = ref(obs.double().cpu())
        out = snap_p.heads(obs)
    assert [w.data_ptr() for w in snap_p.wp] == ptrs
    assert float((out.double().cpu() - torch.cat([lg64, v64[:, None]], 1)).abs().max()) < 2e-4 * scale


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_mlp_gemm_x3_group_matches_its_single_launches(layout):
    """brl_mlp_gemm_x3_group: up to 8 bf16x3 products in ONE launch (the DeepMind step's weight gradients, layout TN: three 1024 x 1024 +
    one 1024 x 480 at K = 1024, here with edge-tile shapes and other chunk counts beside them): the SAME bits as brl_mlp_gemm_x3 without a
    workspace product by product (the same tile code, one K slice), each within the exact kernel's bound of float64; 9 products, a
    misaligned operand are refused."""
    import ctypes as C
    from brl_amd import _capi
    L = _capi.lib()
    g = torch.Generator(device="cuda").manual_seed(177 + layout)
    r = lambda *sh: (torch.rand(sh, device="cuda", generator=g) * 2 - 1)  # noqa: E731
    akc, bkc = layout != 2, layout == 0
    shapes = [(1024, 1024, 1024)] * 3 + [(1024, 480, 1024), (200, 680, 992), (64, 36, 64), (4, 4, 32)]
    As = [r(M, K) if akc else r(K, M) for M, N, K in shapes]
    Bs = [r(N, K) if bkc else r(K, N) for M, N, K in shapes]
    Cs = [torch.full((M, N), float("nan"), device="cuda") for M, N, K in shapes]
    n = len(shapes)
    vp, i64 = C.c_void_p * n, C.c_int64 * n
    s = torch.cuda.current_stream().cuda_stream
    args = (vp(*[t.data_ptr() for t in As]), i64(*[t.stride(0) for t in As]), vp(*[t.data_ptr() for t in Bs]), i64(*[t.stride(0) for t in Bs]),
            vp(*[t.data_ptr() for t in Cs]), i64(*[sh[1] for sh in shapes]), i64(*[sh[0] for sh in shapes]), i64(*[sh[1] for sh in shapes]),
            i64(*[sh[2] for sh in shapes]))
    _capi.check(L.brl_mlp_gemm_x3_group(0, layout, n, *args, s))
    for A, Bm, Cg, (M, N, K) in zip(As, Bs, Cs, shapes):
        ref = (A.double() if akc else A.double().t()) @ (Bm.double().t() if bkc else Bm.double())
        assert float((Cg.double() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())) * (K / 1024 + 1) ** 0.5
        one = torch.full((M, N), float("nan"), device="cuda")
        _capi.check(L.brl_mlp_gemm_x3(0, layout, 0, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), one.data_ptr(), N, M, N, K, 0,
                                      None, None, 0, None, None, 0, s))
        assert torch.equal(one, Cg)
    with pytest.raises(_capi.BrlError):
        _capi.check(L.brl_mlp_gemm_x3_group(0, layout, 9, *args, s))
    off = vp(*([As[0].data_ptr() + 4] + [t.data_ptr() for t in As[1:]]))
    with pytest.raises(_capi.BrlError):
        _capi.check(L.brl_mlp_gemm_x3_group(0, layout, n, off, *args[1:], s))


def test_mlp_gemm_rejects_what_it_cannot_do():
    from brl_amd import _capi
    L = _capi.lib()
    x = torch.zeros(64, 64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for bad in ((0, 1, 64, 62, 64), (0, 0, 64, 64, 66), (3, 0, 64, 64, 64), (1, 1, 64, 64, 64)):   # n % 4, k % 4, layout, epilogue / layout pair
        lay, epi, m, n, k = bad
        assert L.brl_mlp_gemm(0, lay, epi, x.data_ptr(), 64, x.data_ptr(), 64, x.data_ptr(), 64, m, n, k, 0, x.data_ptr(), None, 0, None,
                              None, s) == -1
        assert b"bad argument" in L.brl_last_error()


@pytest.mark.parametrize("activation,model,n,m", [("relu", "DeepMind", 1500, 256), ("relu", "DeepMind", 700, 513),
                                                  ("tanh", "DeepMind", 300, 37), ("relu", "DeepMind_6", 1100, 1024)])
def test_mlp_forward_rows_matches_float64(activation, model, n, m):
    """brl_mlp_forward_rows (the evaluators' small-batch forward: cast + layers + heads + scatter in one call) against the
    module in float64 on the host, and against the oracle shim's restatement of the same entry point; rows that are not
    selected keep what they held (src/models.py:23-33)."""
    import ctypes as C
    from brl_amd import _capi
    from brl_amd.evaluation import _Forward
    from brl_amd.models import make_forward_pass
    fp = make_forward_pass(activation, model)
    net = fp.init(7, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(n + m)
    obs = torch.rand(n, 480, device="cuda", generator=g) < 0.1
    rows = torch.randperm(n, device="cuda", generator=g)[:m].contiguous() if m <= n else None
    out = torch.full((n, 40), 123.0, device="cuda")
    fwd = _Forward(fp, net)
    assert fwd.ref is not None
    fwd.rows(obs, rows, m, out, None)
    torch.cuda.synchronize()
    net64 = fp.init(7).double()
    with torch.no_grad():
        logits, value = net64(obs[rows].cpu().double())
    got = out[rows].cpu().double()
    scale = max(1.0, float(logits.abs().max()))
    assert float((got[:, :38] - logits).abs().max()) < 2e-4 * scale
    assert float((got[:, 38] - value).abs().max()) < 2e-4 * scale
    untouched = torch.ones(n, dtype=torch.bool)
    untouched[rows.cpu()] = False
    assert bool((out.cpu()[untouched] == 123.0).all()) and bool((out[:, 39] == 123.0).all())
    # ... and the oracle shim's restatement of the entry point on the host (same struct, host pointers)
    import os
    import oracle as oracle_pkg
    oracle_pkg.build()
    from oracle.binding import shim_path
    shim = C.CDLL(shim_path())
    cpu = fp.init(7)
    r = _capi.MlpRef()
    r.nlayers, r.act, r.in_features, r.hidden = len(cpu.body), 0 if activation == "relu" else 1, 480, 1024
    for i, lin in enumerate(cpu.body):
        r.w[i], r.b[i] = lin.weight.data_ptr(), lin.bias.data_ptr()
    r.actor_w, r.actor_b, r.critic_w, r.critic_b = (cpu.actor.weight.data_ptr(), cpu.actor.bias.data_ptr(),
                                                    cpu.critic.weight.data_ptr(), cpu.critic.bias.data_ptr())
    k = min(m, 24)   # (a few rows: the shim is a plain triple loop)
    obs_h, rows_h = obs.cpu().to(torch.uint8).contiguous(), rows[:k].cpu().contiguous()
    scratch, out_h = torch.empty(k * (480 + 2048)), torch.full((n, 40), 123.0)
    shim.brl_mlp_forward_rows.argtypes = [C.c_int, C.POINTER(_capi.MlpRef)] + [C.c_void_p] * 2 + [C.c_int64, C.c_void_p, C.c_int64,
                                                                                               C.c_void_p, C.c_int64, C.c_void_p]
    assert shim.brl_mlp_forward_rows(0, C.byref(r), obs_h.data_ptr(), rows_h.data_ptr(), k, scratch.data_ptr(), scratch.numel(),
                                     out_h.data_ptr(), 40, None) == 0
    assert float((out_h[rows_h][:, :39].double() - got[:k, :39]).abs().max()) < 2e-4 * scale


def test_mlp_forward_rows_takes_parameters_inside_the_fused_update_buffers():
    """The learner's parameters are views of FusedMinibatch's flat buffers (critic.bias sits 8 bytes off a 16-byte boundary there):
    the evaluators' one-call forward must still apply to them, and read the live values."""
    import copy
    from brl_amd.evaluation import _Forward
    from brl_amd.models import make_forward_pass
    from brl_amd.train import DEFAULTS
    from brl_amd.update import FusedMinibatch, make_optimizer
    fp = make_forward_pass("relu", "DeepMind")
    net = fp.init(11, device="cuda")
    cfg = dict(DEFAULTS, num_envs=256, num_steps=4, minibatch_size=256, update_epochs=1)
    fm = FusedMinibatch(cfg, net, make_optimizer(cfg, net)["opt"], 256, torch.device("cuda"))
    assert net.critic.bias.data_ptr() == fm.P[fm.views[net.critic.bias]].data_ptr() and net.critic.bias.data_ptr() % 16 != 0
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))   # (in place: the views stay views)
    fwd = _Forward(fp, net)
    assert fwd.ref is not None
    n, m = 900, 300
    g = torch.Generator(device="cuda").manual_seed(5)
    obs = torch.rand(n, 480, device="cuda", generator=g) < 0.1
    rows = torch.randperm(n, device="cuda", generator=g)[:m].contiguous()
    out = torch.zeros((n, 39), device="cuda")
    fwd.rows(obs, rows, m, out, None)
    net64 = copy.deepcopy(net).cpu().double()
    with torch.no_grad():
        logits, value = net64(obs[rows].cpu().double())
    got = out[rows].cpu().double()
    scale = max(1.0, float(logits.abs().max()))
    assert float((got[:, :38] - logits).abs().max()) < 2e-4 * scale and float((got[:, 38] - value).abs().max()) < 2e-4 * scale


def test_mlp_forward_rows_rejects_what_it_cannot_do():
    import ctypes as C
    from brl_amd import _capi
    L = _capi.lib()
    x = torch.zeros(1 << 16, device="cuda")   # (every array of the valid call lies inside it: w[0] is 64 x 480 floats)
    r = _capi.MlpRef()
    r.nlayers, r.act, r.in_features, r.hidden = 1, 0, 480, 64
    r.w[0] = r.b[0] = r.actor_w = r.actor_b = r.critic_w = r.critic_b = x.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    args = lambda: (0, C.byref(r), x.data_ptr(), None, 1, x.data_ptr(), x.numel(), x.data_ptr(), 40, s)   # noqa: E731
    assert L.brl_mlp_forward_rows(*args()) == 0
    for field, bad in (("nlayers", 9), ("in_features", 481), ("hidden", 1028), ("hidden", 62), ("act", 2)):
        keep = getattr(r, field)
        setattr(r, field, bad)
        assert L.brl_mlp_forward_rows(*args()) == -1 and b"bad argument" in L.brl_last_error(), field
        setattr(r, field, keep)
    assert L.brl_mlp_forward_rows(0, C.byref(r), x.data_ptr(), None, 1, x.data_ptr(), 100, x.data_ptr(), 40, s) == -1   # scratch too small
    torch.cuda.synchronize()


def test_longest_auction_319_calls(env, oracle, dds):
    """Maximum size of the domain: the 319-call auction fills every history nibble and the 9-bit turn counter."""
    from tests.test_oracle_kat import longest_auction
    calls = longest_auction()
    n = 8
    hands = np.stack([oracle.key_to_hand(dds["keys"][i]) for i in range(n)])
    tricks = dds["tricks"][:n].reshape(n, 20)
    st = env.init_from_deals(hands, 3, True, True, [1, 3, 0, 2], tricks)
    ref = oracle.init_explicit(hands, 3, 1, 1, [1, 3, 0, 2], tricks)
    for i, a in enumerate(calls):
        st = env.step(st, torch.full((n,), a, dtype=torch.int32), inplace=True)
        oracle.step(ref, np.full(n, a, np.int32))
        if i % 40 == 0 or i > 312:
            assert_state_equal(st, ref, where=f"longest auction call {i}")
    assert ref["terminated"].all() and int(ref["turn"][0]) == 318


def test_rollout_regression_vectors_gpu(env):
    """The HIP rollout reproduces the committed SHA-256 regression vectors without the oracle in the loop."""
    import hashlib, json
    import brl_amd
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_regression.json")
    for rec in json.load(open(here)):
        c = rec["case"]
        roll = brl_amd.make_random_roll_out({"num_steps": c["T"], "substeps": c["substeps"]}, env)
        st = env.init(c["seed"], num_envs=c["n"])
        rs, traj = roll((None, None, st, None, 0, 0))
        torch.cuda.synchronize()
        assert int(rs[4].item()) == rec["terminated_count"]
        for k, want in rec["sha256"].items():
            assert hashlib.sha256(np.ascontiguousarray(to_np(getattr(traj, k))).tobytes()).hexdigest() == want, k
