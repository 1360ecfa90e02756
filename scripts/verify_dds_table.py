#!/usr/bin/env python3
"""verify_dds_table.py — is a pgx ``dds_results/*.npy`` file packed the way brl_amd assumes?

    python scripts/verify_dds_table.py dds_results/train_000.npy [--max-rows 100000] [--json]

The packing brl_amd (and its oracle) restate from pgx 1.4.0 could not be checked against pgx in the build container
([RECALL] items of DESIGN.md §5): array (2, L, 4) int32 = (keys, values);
  key    one word per suit S,H,D,C; 13 base-4 digits, most significant first, ranks A,2,..,K; digit = owner seat N,E,S,W
  value  one word per declarer seat N,E,S,W; 5 hex digits, most significant first = tricks in C,D,H,S,NT
This tool tests every consequence of that packing that a WRONG packing would break, using only bridge facts, and
searches the alternatives, so a maintainer with a real file gets a yes/no in one command (ppo.py:297-308 loads them):

  structure   every key word < 4^13 and the four words give each seat exactly 13 cards; every value word < 16^5 and
              every digit <= 13.
  seats       declarers N and S (and E, W) take the same tricks in most strains, and N's tricks + E's tricks in the
              same strain are 13 +- 1 almost always (the fixture from wb5/dataset_for_vs_wb5.json: 74 % of deals have
              N == S in all five strains; |N + E - 13| <= 1 for 92 % of (deal, strain) pairs).  Any other assignment
              of the four value words to seats breaks one of the two.
  declarer    the two symmetric seat tests cannot tell a value word from its partner's; what can: the hand that
              holds more high cards takes (on average) more tricks AS DECLARER than its partner would, because the
              opening lead comes up to its honours — tricks(word s) - tricks(word s+2) must correlate POSITIVELY with
              hcp(seat s) - hcp(seat s+2) (+0.15 in the fixture, 7 sigma at 1000 deals); swapped partners give -0.15.
  strains     a side's tricks in a suit contract grow with its combined length in THAT suit: the value digit of suit X
              must correlate best with the length the key words give for suit X; the no-trump digit is the one that is
              almost never above the best suit (98.7 % in the fixture) and correlates with high-card points instead.
              All 120 assignments of the five digits are scored; the assumed one must win.
  ranks       high-card points computed with the assumed rank order (A first, then 2..K) must correlate with no-trump
              tricks better than with any cyclic shift of the rank digits.
What it CANNOT see: a relabelling applied consistently to keys AND values (e.g. clubs <-> spades in both) — double-dummy
results are symmetric under it; only the scoring table (minors 20, majors 30) is not.  The JSON fixture the tests use
names suits explicitly, so the repo's own tables are safe from it.

Exit code 0 = every check passed, 1 = something is off (the report says which alternative fits better)."""
from __future__ import annotations

import argparse
import itertools
import json
import sys

import numpy as np

SEATS = "NESW"
STRAINS = ["C", "D", "H", "S", "NT"]
KEY_SUITS = "SHDC"                      # assumed order of the four key words
HCP_ASSUMED = np.array([4] + [0] * 9 + [1, 2, 3], np.float64)   # ranks A,2,..,T,J,Q,K


def decode_keys(keys):
    """-> owner [L,4 words,13 digits] (seat 0..3), msd first"""
    k = keys.astype(np.int64)[:, :, None]
    return (k >> (2 * np.arange(12, -1, -1))) & 3


def decode_values(values):
    """-> digits [L,4 words,5 digits], msd first"""
    v = values.astype(np.int64)[:, :, None]
    return (v >> (4 * np.arange(4, -1, -1))) & 15


def corr(a, b):
    a = a - a.mean()
    b = b - b.mean()
    d = np.sqrt((a * a).sum() * (b * b).sum())
    return float((a * b).sum() / d) if d > 0 else 0.0


def check_structure(keys, values):
    owner = decode_keys(keys)
    counts = np.stack([(owner == s).sum(axis=(1, 2)) for s in range(4)], 1)
    dig = decode_values(values)
    return {
        "key_words_below_4^13": bool(((keys >= 0) & (keys.astype(np.int64) < 4 ** 13)).all()),
        "13_cards_per_seat": float((counts == 13).all(1).mean()),
        "value_words_below_16^5": bool(((values >= 0) & (values.astype(np.int64) < 16 ** 5)).all()),
        "digits_at_most_13": float((dig <= 13).all(axis=(1, 2)).mean()),
    }


def seat_score(dig, order):
    """order[s] = which value word holds declarer seat s.  Higher is better."""
    t = dig[:, list(order), :].astype(np.int64)
    same = ((t[:, 0] == t[:, 2]).mean() + (t[:, 1] == t[:, 3]).mean()) / 2
    compl = ((np.abs(t[:, 0] + t[:, 1] - 13) <= 1).mean() + (np.abs(t[:, 2] + t[:, 3] - 13) <= 1).mean()) / 2
    return float(same + compl)


def strain_scores(owner, dig):
    """corr[d, w]: correlation over (deal, side) of the side's tricks in value digit d with its combined length in key
    word w; nt[d]: how often digit d is <= the maximum of the other four (per declarer)."""
    length = np.stack([(owner == s).sum(2) for s in range(4)], 1).astype(np.float64)    # [L, seat, word]
    side_len = np.concatenate([length[:, 0] + length[:, 2], length[:, 1] + length[:, 3]])  # [2L, word]
    side_tr = np.concatenate([(dig[:, 0] + dig[:, 2]) / 2.0, (dig[:, 1] + dig[:, 3]) / 2.0])  # [2L, digit]
    c = np.array([[corr(side_tr[:, d], side_len[:, w]) for w in range(4)] for d in range(5)])
    flat = dig.reshape(-1, 5)
    nt = np.array([(flat[:, d] <= np.delete(flat, d, axis=1).max(1)).mean() for d in range(5)])
    return c, nt


def best_strain_assignment(c, nt):
    """Score every assignment of the 5 digits to (C, D, H, S, NT); suit X lives in key word KEY_SUITS.index(X)."""
    word_of = {s: KEY_SUITS.index(s) for s in "CDHS"}
    rows = []
    for perm in itertools.permutations(range(5)):            # perm[i] = digit position that holds STRAINS[i]
        s = sum(c[perm[i], word_of[STRAINS[i]]] for i in range(4)) + nt[perm[4]]
        rows.append((s, perm))
    rows.sort(reverse=True)
    return rows


def declarer_score(owner, dig):
    """(correlation, z): tricks(word s) - tricks(word s+2) against hcp(seat s) - hcp(seat s+2), both partnerships"""
    hcp = np.stack([((owner == s) * HCP_ASSUMED).sum(axis=(1, 2)) for s in range(4)], 1)
    dt = np.concatenate([(dig[:, 0].astype(np.float64) - dig[:, 2]).sum(1), (dig[:, 1].astype(np.float64) - dig[:, 3]).sum(1)])
    dh = np.concatenate([hcp[:, 0] - hcp[:, 2], hcp[:, 1] - hcp[:, 3]])
    r = corr(dt, dh)
    return r, r * np.sqrt(len(dt))


def rank_scores(owner, dig):
    """correlation of a side's no-trump tricks with its high-card points under each cyclic shift of the rank digits"""
    side_nt = np.concatenate([(dig[:, 0, 4] + dig[:, 2, 4]) / 2.0, (dig[:, 1, 4] + dig[:, 3, 4]) / 2.0])
    out = []
    for shift in range(13):
        w = np.roll(HCP_ASSUMED, shift)
        hcp = np.stack([((owner == s) * w).sum(axis=(1, 2)) for s in range(4)], 1)
        side = np.concatenate([hcp[:, 0] + hcp[:, 2], hcp[:, 1] + hcp[:, 3]])
        out.append(corr(side_nt, side))
    return out


def verify(keys, values):
    keys = np.ascontiguousarray(keys, np.int32).reshape(-1, 4)
    values = np.ascontiguousarray(values, np.int32).reshape(-1, 4)
    rep = {"rows": int(len(keys)), "structure": check_structure(keys, values)}
    owner, dig = decode_keys(keys), decode_values(values)
    seats = sorted(((seat_score(dig, o), o) for o in itertools.permutations(range(4))), reverse=True)
    id_seat = seat_score(dig, (0, 1, 2, 3))
    # (N,E,S,W), (S,W,N,E), (E,S,W,N)... score alike under the two symmetric tests (partners swap, sides swap): the
    # assumed order must be among the best, and word 0 / word 2 must be partners
    rep["seats"] = {"assumed_score": id_seat, "best_score": seats[0][0], "best_order": list(seats[0][1]),
                    "ok": bool(id_seat >= seats[0][0] - 1e-9)}
    r, z = declarer_score(owner, dig)
    rep["declarer"] = {"corr_trick_diff_x_hcp_diff": round(r, 4), "z": round(float(z), 2), "ok": bool(z > 3.0)}
    c, nt = strain_scores(owner, dig)
    ranked = best_strain_assignment(c, nt)
    ident = tuple(range(5))
    id_score = next(s for s, p in ranked if p == ident)
    rep["strains"] = {"corr_digit_x_keyword": np.round(c, 3).tolist(), "digit_le_max_of_others": np.round(nt, 4).tolist(),
                      "assumed_score": float(id_score), "best_score": float(ranked[0][0]),
                      "best_digit_of_strain": {STRAINS[i]: int(ranked[0][1][i]) for i in range(5)},
                      "runner_up_score": float(ranked[1][0]), "ok": bool(ranked[0][1] == ident)}
    rk = rank_scores(owner, dig)
    rep["ranks"] = {"corr_nt_tricks_hcp_by_shift": np.round(rk, 3).tolist(), "ok": bool(int(np.argmax(rk)) == 0)}
    st = rep["structure"]
    rep["ok"] = bool(st["key_words_below_4^13"] and st["value_words_below_16^5"] and st["13_cards_per_seat"] == 1.0
                     and st["digits_at_most_13"] == 1.0 and rep["seats"]["ok"] and rep["declarer"]["ok"] and rep["strains"]["ok"]
                     and rep["ranks"]["ok"])
    return rep


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("path")
    ap.add_argument("--max-rows", type=int, default=100_000)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    arr = np.load(a.path)
    if arr.ndim != 3 or arr.shape[0] != 2 or arr.shape[2] != 4:
        print(f"{a.path}: expected shape (2, L, 4), got {arr.shape}")
        sys.exit(1)
    rep = verify(arr[0][: a.max_rows], arr[1][: a.max_rows])
    if a.json:
        print(json.dumps(rep))
    else:
        print(f"{a.path}: {rep['rows']} rows")
        for k in ("structure", "seats", "declarer", "strains", "ranks"):
            print(f"  {k}: {json.dumps(rep[k])}")
        print("PACKING OK — matches what brl_amd.load_dds_table assumes" if rep["ok"] else
              "PACKING MISMATCH — see the best_* entries above; adapt brl_amd/bridge_bidding.py:load_dds_table")
    sys.exit(0 if rep["ok"] else 1)


if __name__ == "__main__":
    main()
