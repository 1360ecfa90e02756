"""CPU oracle package — TEST INFRASTRUCTURE ONLY (parity unpinned vs pgx, see bridge_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  ``brl_amd`` never does: the product path fails loudly when the HIP
extension is missing instead of falling back to anything in here.

* ``bridge_oracle.c``  — plain-C restatement of the hot path (the oracle proper).
* ``binding.py``       — ctypes/numpy view of that C library.
* ``brl_shim.c``       — the same oracle behind the C symbols of include/brl_hip.h (host pointers), so that a scenario
  written against the C-ABI runs on either library (tests/test_abi_either_library.py).
* ``eval_stats.py``    — numpy restatement of make_evaluate's statistics (src/evaluation.py:583-1031).
* ``pyref.py``         — a second, independent pure-Python/numpy restatement of
  env.step/observe (history-based, written after wb5/utils.py:15-52) used to cross-check
  the C oracle on small cases.
"""
from .binding import Oracle, STATE_DTYPE, TABLE_INFO_DTYPE, build, lib_path  # noqa: F401
