"""Regenerates tests/golden/wb5_dds_1000.npz from the reference's data file.

Run in the build container only (reads /root/reference, which never travels):
    python tests/golden/make_fixtures.py

Source: /root/reference/wb5/dataset_for_vs_wb5.json — 1000 deals with dealer,
vulnerability and double-dummy tricks per (declarer, strain)  (SURVEY §8c item 5).
Output arrays (DATA only — no reference source text):
    keys    int32 [1000,4]  pgx LUT key packing: one word per suit S,H,D,C, 13 base-4 digits
                            (msd first) = owner seat (N,E,S,W = 0..3) of card suit*13+rank,
                            rank order A,2,..,K   (packing as in wb5/vis_pgx.py:13-24)
    values  int32 [1000,4]  one word per declarer seat N,E,S,W, 5 hex digits (msd first)
                            = tricks in C,D,H,S,NT
    tricks  uint8 [1000,4,5] the same, unpacked [declarer][C,D,H,S,NT]
    dealer  int32 [1000]     N,E,S,W = 0..3
    vul_ns, vul_ew uint8 [1000]
    board_id int64 [1000]
"""
import json
import os

import numpy as np

SRC = "/root/reference/wb5/dataset_for_vs_wb5.json"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wb5_dds_1000.npz")

SEATS = "NESW"
PGX_SUITS = "SHDC"
PGX_RANKS = "A23456789TJQK"
STRAINS = ["C", "D", "H", "S", "NT"]


def main():
    logs = json.load(open(SRC))["logs"]
    n = len(logs)
    keys = np.zeros((n, 4), np.int32)
    values = np.zeros((n, 4), np.int32)
    tricks = np.zeros((n, 4, 5), np.uint8)
    dealer = np.zeros(n, np.int32)
    vul_ns = np.zeros(n, np.uint8)
    vul_ew = np.zeros(n, np.uint8)
    board_id = np.zeros(n, np.int64)
    for i, b in enumerate(logs):
        owner = np.full(52, -1, np.int64)
        for seat, name in enumerate(SEATS):
            cards = b["deal"][name]
            assert len(cards) == 13
            for c in cards:
                cid = PGX_SUITS.index(c[0]) * 13 + PGX_RANKS.index(c[1])
                assert owner[cid] == -1
                owner[cid] = seat
        assert (owner >= 0).all()
        for s in range(4):
            k = 0
            for j in range(13):
                k = k * 4 + int(owner[s * 13 + j])
            keys[i, s] = k
        for seat, name in enumerate(SEATS):
            v = 0
            for d, st in enumerate(STRAINS):
                t = int(b["dda"][name][st])
                assert 0 <= t <= 13
                tricks[i, seat, d] = t
                v = v * 16 + t
            values[i, seat] = v
        dealer[i] = SEATS.index(b["dealer"])
        vul = b["vulnerability"]
        assert vul in ("None", "NS", "EW", "Both"), vul
        vul_ns[i] = vul in ("NS", "Both")
        vul_ew[i] = vul in ("EW", "Both")
        board_id[i] = b["board_id"]
    np.savez_compressed(OUT, keys=keys, values=values, tricks=tricks, dealer=dealer,
                        vul_ns=vul_ns, vul_ew=vul_ew, board_id=board_id)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
