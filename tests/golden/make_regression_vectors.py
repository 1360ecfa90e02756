"""Regenerates tests/golden/rollout_regression.json — SHA-256 of the oracle's rollout outputs on fixed
seeds.  These are REGRESSION vectors of this repo's own oracle (not reference outputs: the reference
cannot run here); they pin today's behaviour so that a later change to the oracle or the kernels that
alters any byte is caught by both `-m "not gpu"` (oracle) and `-m gpu` (HIP) suites.
    python tests/golden/make_regression_vectors.py
"""
import hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import Oracle

CASES = [dict(n=512, T=32, seed=2024, substeps=1), dict(n=300, T=16, seed=7, substeps=4)]


def digest(out):
    return {k: hashlib.sha256(np.ascontiguousarray(out[k]).tobytes()).hexdigest()
            for k in ("obs", "legal_action_mask", "action", "done", "reward", "log_prob", "value")}


def main():
    d = np.load(os.path.join(HERE, "wb5_dds_1000.npz"))
    orc = Oracle(d["keys"], d["values"])
    res = []
    for c in CASES:
        st = orc.init_random(c["n"], seed=c["seed"])
        out = orc.rollout_random(st, c["T"], seed=c["seed"], substeps=c["substeps"])
        res.append(dict(case=c, terminated_count=out["terminated_count"], sha256=digest(out)))
    json.dump(res, open(os.path.join(HERE, "rollout_regression.json"), "w"), indent=1)
    print("wrote", len(res), "cases")


if __name__ == "__main__":
    main()
