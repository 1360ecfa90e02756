"""scripts/verify_dds_table.py discriminates: on the 1000-deal fixture (data derived from the reference's
wb5/dataset_for_vs_wb5.json, packed as pgx is believed to pack it) the assumed packing passes every check, and every
one of the 119 other strain-digit orders, the 23 other seat orders of the value words, the other key-word orders and
shifted rank digits are REJECTED — with the applied strain permutation recovered exactly.  So running the tool on a real
``dds_results/*.npy`` answers the [RECALL] questions of DESIGN.md §5 (ppo.py:297-308)."""
import itertools
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from verify_dds_table import STRAINS, decode_keys, decode_values, verify  # noqa: E402


def pack_values(dig):
    return (dig.astype(np.int64) * (16 ** np.arange(4, -1, -1))).sum(-1).astype(np.int32)


def pack_keys(owner):
    return (owner.astype(np.int64) * (4 ** np.arange(12, -1, -1))).sum(-1).astype(np.int32)


def test_assumed_packing_passes(dds):
    rep = verify(dds["keys"], dds["values"])
    assert rep["ok"], rep
    assert rep["strains"]["best_digit_of_strain"] == {s: i for i, s in enumerate(STRAINS)}
    assert rep["strains"]["best_score"] - rep["strains"]["runner_up_score"] > 0.3
    assert rep["declarer"]["z"] > 5 and rep["ranks"]["corr_nt_tricks_hcp_by_shift"][0] > 0.8


def test_every_strain_permutation_is_identified(dds):
    dig = decode_values(dds["values"])
    for perm in itertools.permutations(range(5)):
        vals = pack_values(dig[:, :, list(perm)])          # new digit j holds the old digit perm[j]
        rep = verify(dds["keys"], vals)
        want = {s: perm.index(i) for i, s in enumerate(STRAINS)}
        assert rep["strains"]["best_digit_of_strain"] == want, perm
        assert rep["ok"] == (perm == tuple(range(5))), perm


def test_every_seat_order_but_the_assumed_one_is_rejected(dds):
    for order in itertools.permutations(range(4)):
        rep = verify(dds["keys"], dds["values"][:, list(order)])
        assert rep["ok"] == (order == (0, 1, 2, 3)), (order, rep["seats"], rep["declarer"])


def test_key_word_order_and_rank_shift_are_rejected(dds):
    owner = decode_keys(dds["keys"])
    for order in itertools.permutations(range(4)):
        if order != (0, 1, 2, 3):
            assert not verify(pack_keys(owner[:, list(order)]), dds["values"])["ok"], order
    for shift in range(1, 13):
        assert not verify(pack_keys(np.roll(owner, shift, axis=2)), dds["values"])["ranks"]["ok"], shift
    bad = dds["keys"].copy()
    bad[0, 0] ^= 1                                           # one card changes hands: 14 / 12 cards
    assert verify(bad, dds["values"])["structure"]["13_cards_per_seat"] < 1.0


def test_command_line(tmp_path, dds):
    good, bad = tmp_path / "good.npy", tmp_path / "bad.npy"
    np.save(good, np.stack([dds["keys"], dds["values"]]))
    np.save(bad, np.stack([dds["keys"], pack_values(decode_values(dds["values"])[:, :, ::-1])]))   # NT first
    tool = os.path.join(ROOT, "scripts", "verify_dds_table.py")
    r = subprocess.run([sys.executable, tool, str(good)], capture_output=True, text=True)
    assert r.returncode == 0 and "PACKING OK" in r.stdout
    r = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "PACKING MISMATCH" in r.stdout
