"""Second, independent restatement of env.step / observe in pure Python.  TEST INFRASTRUCTURE ONLY.

Written from the *rules*  and the observation spec of wb5/utils.py:15-52,
not from bridge_oracle.c: the legal mask here is DERIVED from (last bid, doubling state,
whose turn) instead of being carried as an array, and the declarer is found by scanning
the auction instead of a first-denomination table.  Small cases only (pure-Python loops).
"""
from __future__ import annotations

import numpy as np

PASS, X, XX, BID0 = 0, 1, 2, 3  # src/duplicate.py:9-12


def card_to_obs_index(card: int) -> int:
    """pgx card (suit S,H,D,C x rank A,2..K) -> obs index rank*4+suit (C,D,H,S x 2..A), wb5/utils.py:18-19."""
    suit, rank = divmod(card, 13)
    return ((rank - 1) % 13) * 4 + (3 - suit)


def score(strain: int, level: int, vul: bool, dbl: int, tricks: int) -> int:
    """Duplicate score for declarer's side; dbl in {0,1,2}. Public laws of duplicate bridge."""
    need = level + 6
    if tricks < need:
        u = need - tricks
        if dbl == 0:
            return -u * (100 if vul else 50)
        if vul:
            table = [200 + 300 * i for i in range(13)]
        else:
            table = [100, 300, 500] + [800 + 300 * i for i in range(10)]
        return -table[u - 1] * (2 if dbl == 2 else 1)
    trick_value = 20 if strain in (0, 1) else 30
    points = (trick_value * level + (10 if strain == 4 else 0)) * (1, 2, 4)[dbl]
    s = points + ((500 if vul else 300) if points >= 100 else 50)
    if level == 6:
        s += 750 if vul else 500
    if level == 7:
        s += 1500 if vul else 1000
    s += (0, 50, 100)[dbl]
    over = tricks - need
    if dbl == 0:
        s += over * trick_value
    else:
        s += over * (100 if not vul else 200) * dbl
    return s


class PyTable:
    """One table. `calls` is the auction so far; everything else is recomputed from it."""

    def __init__(self, hand, dealer, vul_ns, vul_ew, shuffled, tricks):
        self.hand = [int(c) for c in hand]  # 13 cards per seat N,E,S,W
        self.dealer = int(dealer)
        self.vul = (bool(vul_ns), bool(vul_ew))
        self.shuffled = [int(p) for p in shuffled]  # seat -> player id
        self.tricks = np.asarray(tricks, dtype=np.int64).reshape(4, 5)
        self.calls: list[int] = []
        self.terminated = False
        self.rewards = [0.0] * 4

    # -- auction facts, recomputed from the call list -------------------------------
    def seat_of_call(self, i):
        return (self.dealer + i) % 4

    def _facts(self):
        last_bid, bidder_seat, dbl, passes = -1, -1, 0, 0
        for i, c in enumerate(self.calls):
            if c >= BID0:
                last_bid, bidder_seat, dbl, passes = c - BID0, self.seat_of_call(i), 0, 0
            elif c == PASS:
                passes += 1
            else:
                dbl, passes = (1 if c == X else 2), 0
        return last_bid, bidder_seat, dbl, passes

    @property
    def seat_to_act(self):
        # the seat does not advance on the terminating call
        n = len(self.calls) - (1 if self.terminated else 0)
        return (self.dealer + n) % 4

    @property
    def current_player(self):
        return self.shuffled[self.seat_to_act]

    def legal_mask(self):
        m = np.zeros(38, dtype=np.uint8)
        if self.terminated:
            m[:] = 1
            return m
        last_bid, bidder_seat, dbl, _ = self._facts()
        m[PASS] = 1
        m[BID0 + last_bid + 1:] = 1
        if last_bid >= 0:
            opp = (bidder_seat - self.seat_to_act) % 2 == 1
            m[X] = opp and dbl == 0
            m[XX] = (not opp) and dbl == 1
        return m

    def step(self, action):
        if self.terminated:
            self.rewards = [0.0] * 4
            return
        assert self.legal_mask()[action], "pyref covers legal play only"
        self.calls.append(int(action))
        last_bid, bidder_seat, dbl, passes = self._facts()
        self.rewards = [0.0] * 4
        if (last_bid < 0 and passes == 4) or (last_bid >= 0 and passes == 3):
            self.terminated = True
            if last_bid >= 0:
                strain, level = last_bid % 5, last_bid // 5 + 1
                side = bidder_seat % 2
                declarer = next(
                    self.seat_of_call(i)
                    for i, c in enumerate(self.calls)
                    if c >= BID0 and (c - BID0) % 5 == strain and self.seat_of_call(i) % 2 == side
                )
                sc = score(strain, level, self.vul[side], dbl, int(self.tricks[declarer, strain]))
                for seat in range(4):
                    self.rewards[self.shuffled[seat]] = float(sc if seat % 2 == side else -sc)

    def observe(self, player_id=None):
        """wb5/utils.py:15-52 (convert_vul / convert_history / convert_hand) restated."""
        seat = self.seat_to_act if player_id is None else self.shuffled.index(player_id)
        obs = np.zeros(480, dtype=np.uint8)
        we, they = self.vul[seat % 2], self.vul[1 - seat % 2]
        obs[0:4] = [not we, we, not they, they]
        hist = obs[4:428]
        last = -1
        for i, c in enumerate(self.calls):
            rel = (self.seat_of_call(i) - seat) % 4
            if c >= BID0:
                last = c - BID0
                hist[4 + last * 12 + rel] = 1
            elif c == PASS:
                if last < 0:
                    hist[rel] = 1
            elif c == X:
                hist[4 + last * 12 + 4 + rel] = 1
            else:
                hist[4 + last * 12 + 8 + rel] = 1
        for c in self.hand[seat * 13:(seat + 1) * 13]:
            obs[428 + card_to_obs_index(c)] = 1
        return obs
