"""Host model of the logic wave's chain-friendly auction word (brl_amd/csrc/rollout_common.hpp: fast_step,
fast_from_legacy, fast_to_legacy) against a host model of the packed-scalar transition it replaces
(bridge_device.hpp: lean_random_step).  The device functions are the same arithmetic; the GPU parity suite
checks them end to end (k_rollout_pipe), this test pins the ENCODING: doubling state as a 3-bit code whose
legality is bit e of 0x21, pass count biased by "a bid exists" so that bit 22 means "auction over"."""
import random

M32 = 0xFFFFFFFF


def bits(x, p, n):
    return (x >> p) & ((1 << n) - 1)


def mulhi(a, b):
    return (a * b) >> 32


def legacy_step(sc, sch, u):
    """lean_random_step on the packed scalars (SC_* layout of bridge_device.hpp)."""
    lb1 = bits(sc, 12, 6)
    seat = (bits(sc, 0, 2) + bits(sch, 0, 9)) & 3
    own = ((bits(sc, 18, 2) ^ seat) & 1) ^ 1
    x, xx, has = bits(sc, 20, 1), bits(sc, 21, 1), int(lb1 != 0)
    can_x = has & (own ^ 1) & (x ^ 1) & (xx ^ 1)
    can_xx = has & own & x & (xx ^ 1)
    dbl = can_x | can_xx
    n = 36 - lb1 + dbl
    k = mulhi(u, n)
    a = (1 if can_x else 2) if (dbl and k == 1) else 2 + lb1 + k - dbl
    if k == 0:
        a = 0
    was_term = bits(sc, 25, 1)
    sc &= ~(1 << 25)
    if was_term:
        sch &= ~(1023 << 9)
    is_pass, is_bid, b = a == 0, a >= 3, a - 3
    pas = bits(sc, 22, 3) + 1 if is_pass else 0
    set_dbl = 0 if is_bid else ((1 << 20) if a == 1 else 0) | ((1 << 21) if a == 2 else 0)
    clear = (63 << 12) | (3 << 18) | (1 << 20) | (1 << 21)
    nsc = ((sc & ~clear) | ((b + 1) << 12) | (seat << 18)) if is_bid else (sc | set_dbl)
    nlb1 = b + 1 if is_bid else lb1
    term = int(pas == (3 if nlb1 else 4))
    nsc = (nsc & ~(7 << 22)) | (pas << 22) | (((1 << 25) | (1 << 26)) if term else 0)
    sch += (1 << 9) + (0 if term else 1)
    return nsc & M32, sch & M32, term, a, n


def fast_from_legacy(sc, sch):
    lb1 = bits(sc, 12, 6)
    st = bits(sc, 0, 2) + bits(sch, 0, 9)
    has, x, xx = int(lb1 != 0), bits(sc, 20, 1), bits(sc, 21, 1)
    own = ((bits(sc, 18, 2) ^ st) & 1) ^ 1
    e = ((x + xx) | (own << 2)) if has else 2
    return (st & 0x1FF) | ((35 - lb1) << 9) | (bits(sc, 18, 2) << 15) | (e << 17) | ((bits(sc, 22, 3) + has) << 20)


def fast_to_legacy(d, stw):
    lb1 = 35 - bits(d, 9, 6)
    has, dbl = int(lb1 != 0), bits(d, 17, 2)
    x, xx = has & int(dbl >= 1), has & int(dbl == 2)
    turn = (bits(d, 0, 9) - (stw & 3)) & 0x1FF
    sc = (stw & 0x0A000FFF) | (lb1 << 12) | (bits(d, 15, 2) << 18) | (x << 20) | (xx << 21) | ((bits(d, 20, 3) - has) << 22)
    return sc, turn | (turn << 9)


def fast_step(d, u):
    rem, e = bits(d, 9, 6), bits(d, 17, 3)
    dbl = (0x21 >> e) & 1
    n = rem + dbl + 1
    k = mulhi(u, n)
    kb = k - dbl
    d1 = (d + 1) & M32
    d_pass = ((d1 ^ (4 << 17)) + (1 << 20)) & M32
    d_dbl = ((((d1 + (1 << 17)) ^ (4 << 17)) & ~(7 << 20)) | (1 << 20)) & M32
    d_bid = (d1 & 0x1FF) | ((rem - kb) << 9) | ((d & 3) << 15) | (1 << 20)
    dn = d_pass if k == 0 else d_dbl
    return (d_bid if kb > 0 else dn), n


def draws(rng):
    u = rng.getrandbits(32)
    if rng.random() < 0.5:  # edge draws: first / last legal call, cell boundaries, small values
        u = rng.choice([0, M32, (1 << 32) // 36 + 1, rng.getrandbits(32) >> rng.randrange(8)])
    return u


def test_fast_step_matches_the_packed_scalar_transition():
    rng = random.Random(1)
    checked = 0
    for _ in range(4000):
        stw = rng.randrange(4) | (rng.randrange(4) << 2) | (rng.randrange(256) << 4)  # dealer, vul, seating
        sc, sch = stw, 0
        d = fast_from_legacy(sc, sch)
        for _ in range(400):
            assert fast_to_legacy(d, stw) == (sc, sch)
            u = draws(rng)
            nsc, nsch, term, _, n = legacy_step(sc, sch, u)
            dn, n_fast = fast_step(d, u)
            assert n_fast == n and bits(dn, 22, 1) == term
            if term:
                break
            want = fast_from_legacy(nsc, nsch)
            if bits(nsc, 12, 6) == 0:  # no bid yet: the "own side" bit of the doubling code is a don't-care
                assert (dn & ~(4 << 17)) == (want & ~(4 << 17))
            else:
                assert dn == want
            sc, sch, d = nsc, nsch, dn
            checked += 1
    assert checked > 30000


def test_doubling_code_legality_table():
    # e = dblst | own << 2; X or XX is legal iff (no double yet and the opponents bid) or (doubled and own side bid)
    legal = {e for e in range(8) if (0x21 >> e) & 1}
    assert legal == {0, 5}
