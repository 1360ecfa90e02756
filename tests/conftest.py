import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_runtest_logstart(nodeid, location):
    """BRL_TEST_TRACE=<file>: the test that is about to run, appended and flushed — a run that dies with the process (a GPU
    memory fault aborts it from a runtime thread) still says where it was (scripts/gpu_round.sh sets it)."""
    path = os.environ.get("BRL_TEST_TRACE")
    if path:
        with open(path, "a") as f:
            f.write(nodeid + "\n")


@pytest.fixture(scope="session")
def dds():
    """1000 double-dummy deals derived from the reference's wb5/dataset_for_vs_wb5.json."""
    d = np.load(os.path.join(GOLDEN, "wb5_dds_1000.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def oracle(dds):
    from oracle import Oracle

    return Oracle(dds["keys"], dds["values"])


def synthetic_lut(n, seed=0):
    """Throughput-style LUT: random deals, uniform 0..13 tricks."""
    rng = np.random.default_rng(seed)
    keys = np.zeros((n, 4), np.int32)
    values = np.zeros((n, 4), np.int32)
    for i in range(n):
        owner = np.repeat(np.arange(4), 13)
        rng.shuffle(owner)
        for s in range(4):
            k = 0
            for j in range(13):
                k = k * 4 + int(owner[s * 13 + j])
            keys[i, s] = k
        t = rng.integers(0, 14, size=(4, 5))
        for seat in range(4):
            v = 0
            for d in range(5):
                v = v * 16 + int(t[seat, d])
            values[i, seat] = v
    return keys, values
