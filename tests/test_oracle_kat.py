"""Pins the CPU oracle against every known-answer fragment the reference holds for the
hot path and against the independent pure-Python restatement.
CPU only."""
import numpy as np
import pytest

from oracle import pyref


# ---------------------------------------------------------------- RNG (build-side)
def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kats:
        assert [int(x) for x in oracle.philox(ctr, key)] == want


# ---------------------------------------------------------------- A10 IMP (fully pinned)
IMP_DOCTESTS = [  # /root/reference/src/duplicate.py:20-43
    ([0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]),
    ([0, 0, 0, 0], [100, 100, -100, -100], [3, 3, -3, -3]),
    ([-100, -100, 100, 100], [0, 0, 0, 0], [-3, -3, 3, 3]),
    ([-100, -100, 100, 100], [100, 100, -100, -100], [0, 0, 0, 0]),
    ([-3500, -3500, 3500, 3500], [0, 0, 0, 0], [-23, -23, 23, 23]),
    ([2000, 2000, -2000, -2000], [2000, 2000, -2000, -2000], [24, 24, -24, -24]),
]
IMP_THRESHOLDS = [20, 50, 90, 130, 170, 220, 270, 320, 370, 430, 500, 600, 750, 900, 1100,
                  1300, 1500, 1750, 2000, 2250, 2500, 3000, 3500, 4000]  # src/duplicate.py:46-50


def test_imp_doctests(oracle):
    for a, b, want in IMP_DOCTESTS:
        got = oracle.imp_reward(a, b)
        assert got.dtype == np.float32
        assert got.tolist() == [float(x) for x in want]


def test_imp_thresholds(oracle):
    for i, th in enumerate(IMP_THRESHOLDS):
        assert oracle.imp_reward([th, 0, 0, 0], [0] * 4)[0] == i + 1
        assert oracle.imp_reward([th - 10, 0, 0, 0], [0] * 4)[0] == i
        assert oracle.imp_reward([-th, 0, 0, 0], [0] * 4)[0] == -(i + 1)
    assert oracle.imp_reward([7600, 0, 0, 0], [7600, 0, 0, 0])[0] == 24


# ---------------------------------------------------------------- A4 scoring
def test_score_extremes(oracle):
    # 13 down doubled non-vul = 3500 (src/duplicate.py:36); 13 down redoubled vul = 7600 = reward_scale (ppo.py:174)
    assert oracle.score(4, 7, 0, 1, 0, 0) == -3500
    assert oracle.score(4, 7, 1, 1, 1, 0) == -7600
    assert max(abs(oracle.score(d, l, v, x, xx, t))
               for d in range(5) for l in range(1, 8) for v in (0, 1)
               for (x, xx) in ((0, 0), (1, 0), (1, 1)) for t in range(14)) == 7600


@pytest.mark.parametrize("args,want", [
    # (strain C,D,H,S,NT; level; vul; X; XX; tricks) -> score; laws of duplicate bridge
    ((4, 3, 0, 0, 0, 9), 400), ((4, 3, 1, 0, 0, 9), 600), ((3, 4, 0, 0, 0, 10), 420), ((2, 4, 1, 0, 0, 10), 620),
    ((0, 5, 0, 0, 0, 11), 400), ((1, 5, 1, 0, 0, 11), 600), ((4, 6, 0, 0, 0, 12), 990), ((4, 6, 1, 0, 0, 12), 1440),
    ((3, 6, 0, 0, 0, 12), 980), ((4, 7, 0, 0, 0, 13), 1520), ((4, 7, 1, 0, 0, 13), 2220), ((0, 7, 1, 0, 0, 13), 2140),
    ((4, 1, 0, 0, 0, 7), 90), ((4, 1, 0, 0, 0, 8), 120), ((0, 1, 0, 0, 0, 13), 190), ((2, 2, 0, 0, 0, 8), 110),
    ((2, 2, 1, 1, 0, 8), 670), ((2, 2, 0, 1, 0, 8), 470), ((2, 2, 0, 1, 0, 9), 570), ((2, 2, 1, 1, 0, 9), 870),
    ((4, 1, 1, 1, 1, 7), 760), ((4, 1, 0, 1, 1, 7), 560), ((0, 1, 0, 1, 1, 7), 230), ((0, 1, 0, 1, 0, 7), 140),
    ((4, 1, 0, 1, 1, 8), 760), ((4, 1, 1, 1, 1, 8), 1160), ((4, 7, 1, 1, 1, 13), 2980), ((0, 5, 0, 1, 0, 11), 550),
    ((4, 3, 0, 0, 0, 8), -50), ((4, 3, 1, 0, 0, 6), -300), ((4, 3, 0, 1, 0, 8), -100), ((4, 3, 0, 1, 0, 7), -300),
    ((4, 3, 0, 1, 0, 6), -500), ((4, 3, 0, 1, 0, 5), -800), ((4, 3, 1, 1, 0, 8), -200), ((4, 3, 1, 1, 0, 7), -500),
    ((4, 3, 1, 1, 0, 6), -800), ((4, 3, 1, 1, 1, 8), -400), ((4, 3, 0, 1, 1, 5), -1600), ((4, 3, 1, 1, 1, 5), -2200),
])
def test_score_known_contracts(oracle, args, want):
    assert oracle.score(*args) == want


def test_score_oracle_equals_pyref(oracle):
    for d in range(5):
        for l in range(1, 8):
            for v in (0, 1):
                for dbl in (0, 1, 2):
                    for t in range(14):
                        assert oracle.score(d, l, v, dbl >= 1, dbl == 2, t) == pyref.score(d, l, bool(v), dbl, t)


# ---------------------------------------------------------------- LUT packing
def test_value_packing_docstring_kat(oracle):
    # [RECALL] pgx docstring KAT quoted in SURVEY App. B
    t = oracle.value_to_tricks([4160, 904605, 4160, 904605])
    assert t.tolist() == [0, 1, 0, 4, 0, 13, 12, 13, 9, 13, 0, 1, 0, 4, 0, 13, 12, 13, 9, 13]
    assert oracle.tricks_to_value(t).tolist() == [4160, 904605, 4160, 904605]


def test_key_roundtrip_and_fixture_consistency(oracle, dds):
    for i in range(0, 1000, 7):
        hand = oracle.key_to_hand(dds["keys"][i])
        assert sorted(hand.tolist()) == list(range(52))
        for seat in range(4):
            h = hand[seat * 13:(seat + 1) * 13]
            assert (np.diff(h) > 0).all()
        assert oracle.hand_to_key(hand).tolist() == dds["keys"][i].tolist()
        assert oracle.value_to_tricks(dds["values"][i]).reshape(4, 5).tolist() == dds["tricks"][i].tolist()


def test_card_to_obs_index_is_bijection(oracle):
    idx = [oracle.card_to_obs_index(c) for c in range(52)]
    assert sorted(idx) == list(range(52))
    assert idx == [pyref.card_to_obs_index(c) for c in range(52)]
    # spade ace (pgx 0) -> rank A (12), suit S (3); club two (pgx 40) -> 0
    assert idx[0] == 12 * 4 + 3 and idx[39 + 1] == 0


# ---------------------------------------------------------------- A3 observation, wb5 auction
WB5_AUCTION = [0, 9, 11, 20, 1, 0, 22, 1, 2, 0, 0, 28, 0, 0]  # wb5/utils.py:61-64
WB5_DEALER, WB5_SHUFFLED = 1, [0, 3, 1, 2]  # wb5/utils.py:69-75 ; nobody vulnerable


def spec_obs(dealer, seat, vul_we, vul_they, calls, own_cards_obs_idx):
    """Literal restatement of the layout in wb5/utils.py:15-52 on plain ints."""
    o = np.zeros(480, np.uint8)
    o[0], o[1], o[2], o[3] = (not vul_we), vul_we, (not vul_they), vul_they
    last = 0  # 1-based, 0 = none, as in wb5/utils.py:29
    for i, c in enumerate(calls):
        rel = ((i + dealer) % 4 + (4 - seat)) % 4
        b = {0: 36, 1: 37, 2: 38}.get(c, c - 2)  # bridge_env Bid ints: 1..35 bids, 36 P, 37 X, 38 XX
        if b <= 35:
            last = b
            o[4 + 4 + (b - 1) * 12 + rel] = 1
        elif b == 36:
            if last == 0:
                o[4 + rel] = 1
        elif b == 37:
            o[4 + 4 + (last - 1) * 12 + 4 + rel] = 1
        else:
            o[4 + 4 + (last - 1) * 12 + 8 + rel] = 1
    for k in own_cards_obs_idx:
        o[428 + k] = 1
    return o


def test_wb5_auction_obs_and_mask(oracle, dds):
    hand = oracle.key_to_hand(dds["keys"][0])
    st = oracle.init_explicit(hand, WB5_DEALER, 0, 0, WB5_SHUFFLED, dds["tricks"][0].reshape(20))
    ref = pyref.PyTable(hand, WB5_DEALER, 0, 0, WB5_SHUFFLED, dds["tricks"][0])
    assert st["current_player"][0] == 3  # wb5/utils.py:71
    for i, a in enumerate(WB5_AUCTION):
        assert st["legal_action_mask"][0][a] == 1, f"call {i} of the reference auction must be legal"
        oracle.step(st, [a])
        ref.step(a)
        seat = (WB5_DEALER + i + 1) % 4
        assert st["terminated"][0] == 0
        assert st["current_player"][0] == WB5_SHUFFLED[seat] == ref.current_player
        own = [oracle.card_to_obs_index(c) for c in hand[seat * 13:(seat + 1) * 13]]
        want = spec_obs(WB5_DEALER, seat, False, False, WB5_AUCTION[:i + 1], own)
        assert (st["observation"][0] == want).all()
        assert (ref.observe() == want).all()
        assert (st["legal_action_mask"][0] == ref.legal_mask()).all()
    # after 6C P P the next pass ends the auction: declarer side = whoever bid 6C
    assert st["pass_num"][0] == 2 and st["last_bid"][0] == 25
    m = st["legal_action_mask"][0]
    assert m[0] == 1 and m[1] == 1 and m[2] == 0 and not m[3:29].any() and m[29:].all()


def test_obs_invariants_random_play(oracle, dds):
    rng = np.random.default_rng(1)
    n = 64
    idx = rng.integers(0, 1000, n)
    hands = np.stack([oracle.key_to_hand(dds["keys"][i]) for i in idx])
    sh = np.array([[0, 2, 1, 3]] * n)
    st = oracle.init_explicit(hands, dds["dealer"][idx], dds["vul_ns"][idx], dds["vul_ew"][idx], sh,
                              dds["tricks"][idx].reshape(n, 20))
    for _ in range(60):
        obs = st["observation"]
        assert (obs[:, 428:].sum(1) == 13).all()
        assert (obs[:, 0] + obs[:, 1] == 1).all() and (obs[:, 2] + obs[:, 3] == 1).all()
        m = st["legal_action_mask"]
        live = st["terminated"] == 0
        lb = st["last_bid"]
        for e in np.nonzero(live)[0]:
            assert m[e, 0] == 1
            assert not m[e, 3:3 + lb[e] + 1].any() and m[e, 3 + lb[e] + 1:].all()
            assert not (m[e, 1] and m[e, 2])
        act = np.array([rng.choice(np.nonzero(m[e])[0]) for e in range(n)])
        oracle.step(st, act)


# ---------------------------------------------------------------- oracle == pyref on random auctions
def test_oracle_equals_pyref_random_auctions(oracle, dds):
    rng = np.random.default_rng(7)
    perms = [[0, 2, 1, 3], [1, 3, 0, 2], [2, 0, 3, 1], [3, 1, 2, 0], [0, 3, 1, 2], [2, 1, 3, 0], [1, 2, 0, 3], [3, 0, 2, 1]]
    for trial in range(120):
        i = int(rng.integers(0, 1000))
        hand = oracle.key_to_hand(dds["keys"][i])
        sh = perms[int(rng.integers(0, 8))]
        dealer, vn, ve = int(dds["dealer"][i]), int(dds["vul_ns"][i]), int(dds["vul_ew"][i])
        st = oracle.init_explicit(hand, dealer, vn, ve, sh, dds["tricks"][i].reshape(20))
        ref = pyref.PyTable(hand, dealer, vn, ve, sh, dds["tricks"][i])
        pass_bias = rng.choice([0.0, 0.3, 0.6])
        while True:
            m = st["legal_action_mask"][0]
            assert (m == ref.legal_mask()).all()
            assert (st["observation"][0] == ref.observe()).all()
            assert st["current_player"][0] == ref.current_player
            if st["terminated"][0]:
                break
            legal = np.nonzero(m)[0]
            a = 0 if rng.random() < pass_bias else int(rng.choice(legal))
            oracle.step(st, [a])
            ref.step(a)
            assert st["rewards"][0].tolist() == ref.rewards
            assert bool(st["terminated"][0]) == ref.terminated
        r = st["rewards"][0]
        # +s for one partnership, -s for the other, by PLAYER id; partners are {0,1} and {2,3}
        assert r[0] == r[1] == -r[2] == -r[3]
        # stepping a finished table again: zero reward, nothing moves
        before = st.copy()
        oracle.step(st, [0])
        assert (st["rewards"][0] == 0).all() and st["terminated"][0] == 1
        for f in ("observation", "legal_action_mask", "current_player", "last_bid", "last_bidder", "turn"):
            assert (st[f] == before[f]).all()


def test_pass_out_and_declarer(oracle, dds):
    hand = oracle.key_to_hand(dds["keys"][3])
    tr = dds["tricks"][3]
    st = oracle.init_explicit(hand, 2, 1, 0, [2, 0, 3, 1], tr.reshape(20))
    for _ in range(4):
        oracle.step(st, [0])
    assert st["terminated"][0] == 1 and (st["rewards"][0] == 0).all()
    assert st["last_bid"][0] == -1 and st["last_bidder"][0] == -1 and st["pass_num"][0] == 4  # G13
    # declarer = first of the declaring side to NAME the strain: S(dealer) 1H, W P, N 2H, E P, S P, W P
    st = oracle.init_explicit(hand, 2, 1, 0, [2, 0, 3, 1], tr.reshape(20))
    for a in [3 + 2, 0, 3 + 7, 0, 0, 0]:
        oracle.step(st, [a])
    assert st["terminated"][0] == 1
    assert st["last_bidder"][0] == 2  # player id sitting North
    sc = oracle.score(2, 2, 1, 0, 0, int(tr[2, 2]))  # declarer South (seat 2), NS vulnerable
    want = np.zeros(4, np.float32)
    for seat, p in enumerate([2, 0, 3, 1]):
        want[p] = sc if seat % 2 == 0 else -sc
    assert (st["rewards"][0] == want).all()


# ---------------------------------------------------------------- A11/A12 duplicate
def test_duplicate_seat_swap_kat(oracle, dds):
    # src/duplicate.py:81-86: swapped order [0,2,1,3], dealer 1 -> current_player 2; the source
    # order is therefore [2,0,3,1] (ix = [1,0,3,2], src/duplicate.py:113)
    hand = oracle.key_to_hand(dds["keys"][5])
    st = oracle.init_explicit(hand, 1, 0, 1, [2, 0, 3, 1], dds["tricks"][5].reshape(20))
    oracle.step(st, [35])
    from oracle import TABLE_INFO_DTYPE
    A = np.zeros(1, TABLE_INFO_DTYPE)
    B = np.zeros(1, TABLE_INFO_DTYPE)
    for a in (0, 0, 0):
        oracle.duplicate_step(st, [a], A, B)
    assert A["terminated"][0] == 1 and B["terminated"][0] == 0
    assert A["last_bid"][0] == 32 and A["call_x"][0] == 0
    assert st["shuffled_players"][0].tolist() == [0, 2, 1, 3]
    assert st["dealer"][0] == 1 and st["current_player"][0] == 2 and st["pass_num"][0] == 0
    assert st["terminated"][0] == 0 and (st["rewards"][0] == 0).all()
    assert (st["hand"][0] == hand).all() and st["vul_ew"][0] == 1 and st["vul_ns"][0] == 0
    m = st["legal_action_mask"][0]
    assert m[0] == 1 and m[1] == 0 and m[2] == 0 and m[3:].all()  # src/duplicate.py:116-119
    # table B: pass out -> IMP emitted exactly once (G8)
    for k in range(4):
        oracle.duplicate_step(st, [0], A, B)
        if k < 3:
            assert (st["rewards"][0] == 0).all()
    assert B["terminated"][0] == 1 and (B["rewards"][0] == 0).all()
    want = oracle.imp_reward(A["rewards"][0], B["rewards"][0])
    assert (st["rewards"][0] == want).all()
    oracle.duplicate_step(st, [0], A, B)
    assert (st["rewards"][0] == 0).all() and st["terminated"][0] == 1


# ---------------------------------------------------------------- A5 auto_reset, A7 roll_out
def test_auto_reset_keeps_flags_and_deals_new_board(oracle):
    st = oracle.init_random(8, seed=11)
    first = st.copy()
    for k in range(4):
        oracle.step(st, np.zeros(8, np.int32), autoreset=True, seed=11)
    # four passes: every table passed out, was replaced, but keeps terminated/rewards (src/utils.py:45-55)
    assert (st["terminated"] == 1).all() and (st["rewards"] == 0).all()
    assert (st["board_ctr"] == 1).all() and (st["turn"] == 0).all() and (st["pass_num"] == 0).all()
    assert (st["mask_all"] == 0).all() and (st["legal_action_mask"][:, 1:3] == 0).all()
    assert (st["lut_idx"] != first["lut_idx"]).any()
    oracle.step(st, np.full(8, 3, np.int32), autoreset=True, seed=11)
    assert (st["terminated"] == 0).all() and (st["step_count"] == 1).all()  # src/utils.py:34-43


def test_rollout_semantics(oracle):
    n, T = 16, 40
    st = oracle.init_random(n, seed=5)
    shadow = st.copy()
    out = oracle.rollout_random(st, T, seed=5)
    # replay by hand through step(): pre-step obs/mask (G4), reward of the pre-step actor (G1), done (G2)
    tc = 0
    for t in range(T):
        assert (out["obs"][t] == shadow["observation"]).all()
        assert (out["legal_action_mask"][t] == shadow["legal_action_mask"]).all()
        actor = shadow["current_player"].copy()
        acts = []
        for e in range(n):
            a, nl = oracle.random_action(shadow[e:e + 1], oracle.action_draw(5, e, t))
            acts.append(a)
            assert out["log_prob"][t, e] == np.float32(-np.log(np.float64(nl)))
        assert out["action"][t].tolist() == acts
        oracle.step(shadow, np.array(acts), autoreset=True, seed=5)
        assert (out["done"][t] == shadow["terminated"]).all()
        want_r = shadow["rewards"][np.arange(n), actor] / np.float32(7600.0)
        assert (out["reward"][t] == want_r).all()
        tc += int(shadow["terminated"].sum())
    assert out["terminated_count"] == tc and tc > 0
    assert (out["value"] == 0).all()
    for f in ("observation", "current_player", "board_ctr", "lut_idx", "turn"):
        assert (st[f] == shadow[f]).all()


def test_rollout_macro_step_sums_rewards(oracle):
    n, T = 32, 24
    st = oracle.init_random(n, seed=9)
    shadow = st.copy()
    out = oracle.rollout_random(st, T, seed=9, substeps=4)
    for t in range(T):
        actor = shadow["current_player"].copy()
        rs = np.zeros((n, 4), np.float32)
        term = np.zeros(n, bool)
        for k in range(4):
            acts = [oracle.random_action(shadow[e:e + 1], oracle.action_draw(9, e, 4 * t + k))[0] for e in range(n)]
            if k == 0:
                assert out["action"][t].tolist() == acts
            oracle.step(shadow, np.array(acts), autoreset=True, seed=9)
            rs += shadow["rewards"]
            term |= shadow["terminated"] == 1
        shadow["rewards"] = rs           # src/utils.py:126-128
        shadow["terminated"] = term
        assert (out["done"][t] == term).all()
        assert (out["reward"][t] == rs[np.arange(n), actor] / np.float32(7600)).all()


# ---------------------------------------------------------------- A9 GAE
def test_gae_matches_numpy_f32(oracle):
    rng = np.random.default_rng(3)
    T, N = 32, 50
    done = (rng.random((T, N)) < 0.1).astype(np.uint8)
    value = rng.standard_normal((T, N)).astype(np.float32)
    reward = (rng.standard_normal((T, N)) * 0.1).astype(np.float32)
    last = rng.standard_normal(N).astype(np.float32)
    for gamma, lam in ((1.0, 0.95), (0.99, 0.9)):
        adv, tgt = oracle.gae(done, value, reward, last, gamma, lam)
        g = np.float32(gamma)
        gl = np.float32(gamma * lam)
        gae = np.zeros(N, np.float32)
        nv = last.copy()
        for t in range(T - 1, -1, -1):
            nd = np.float32(1) - done[t].astype(np.float32)
            delta = reward[t] + g * nv * nd - value[t]
            gae = delta + gl * nd * gae
            nv = value[t]
            assert (adv[t] == gae).all()
            assert (tgt[t] == gae + value[t]).all()


def longest_auction():
    """The 319-call auction (pgx's _bidding_history length): P P P, then every bid doubled and redoubled
    with two passes in between, closed by three passes."""
    calls = [0, 0, 0]
    for b in range(35):
        calls += [3 + b, 0, 0, 1, 0, 0, 2]
        calls += [0, 0] if b < 34 else [0, 0, 0]
    return calls


def test_longest_auction_oracle_and_pyref(oracle, dds):
    calls = longest_auction()
    assert len(calls) == 319
    hand = oracle.key_to_hand(dds["keys"][11])
    st = oracle.init_explicit(hand, 3, 1, 1, [1, 3, 0, 2], dds["tricks"][11].reshape(20))
    ref = pyref.PyTable(hand, 3, 1, 1, [1, 3, 0, 2], dds["tricks"][11])
    for i, a in enumerate(calls):
        assert st["terminated"][0] == 0 and st["legal_action_mask"][0][a] == 1, i
        oracle.step(st, [a])
        ref.step(a)
    assert st["terminated"][0] == 1 and ref.terminated and st["turn"][0] == 318
    assert st["last_bid"][0] == 34 and st["call_x"][0] == 1 and st["call_xx"][0] == 1
    assert st["rewards"][0].tolist() == ref.rewards
    obs = st["observation"][0]
    assert (obs == ref.observe()).all()
    assert obs[4:8].sum() == 3 and obs[8:428].reshape(35, 3, 4).sum(-1).min() == 1  # every (bid, event) nibble used once


def test_rollout_regression_vectors(oracle):
    """tests/golden/rollout_regression.json: the oracle still produces the bytes it produced when the
    vectors were committed (self-generated regression vectors, see make_regression_vectors.py)."""
    import json, os
    from tests.golden.make_regression_vectors import digest
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_regression.json")
    for rec in json.load(open(here)):
        c = rec["case"]
        st = oracle.init_random(c["n"], seed=c["seed"])
        out = oracle.rollout_random(st, c["T"], seed=c["seed"], substeps=c["substeps"])
        assert out["terminated_count"] == rec["terminated_count"]
        assert digest(out) == rec["sha256"]
