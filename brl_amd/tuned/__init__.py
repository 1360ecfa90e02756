"""Tuned GEMM solutions for the policy path's shapes on MI355X (torch TunableOp results: for every GEMM shape of the PPO minibatch
step and of the rollout forwards, the fastest rocBLAS / hipBLASLt solution as timed on a gfx950 box of this image — the
library's default heuristic is 3-6 % slower on the M = 1024 products).  The file is keyed by shape and validated by torch
against the running PyTorch / ROCm / hipBLASLt versions and the GPU architecture: on any other stack it is ignored.

    python scripts/tune_gemms.py        # regenerates tunableop_gfx950.csv on a GPU box
"""
from __future__ import annotations

import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_state = {"on": False}


def enable(tuning: bool = False, path: str = PATH) -> bool:
    """Use the committed solutions for the GEMMs issued from now on (lookups only; `tuning`: also time unknown shapes once).
    Returns False when the file is missing, belongs to another PyTorch / ROCm / hipBLASLt stack, or TunableOp is unavailable.
    NOTE: TunableOp is a PROCESS-GLOBAL torch switch — once on, every GEMM of the host process (user code included) is looked
    up in the table; shapes that are not in it keep the default heuristics (lookups only, nothing is timed or written)."""
    import torch
    if not (torch.cuda.is_available() and hasattr(torch.cuda, "tunable")) or (not tuning and not os.path.exists(path)):
        return False
    if _state["on"] and not tuning:
        return True
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(bool(tuning))
    if hasattr(t, "write_file_on_exit"):
        t.write_file_on_exit(bool(tuning))
    if tuning:
        t.set_filename(path)   # (only a tuning run may write: lookups never touch the committed file)
    if os.path.exists(path):
        try:
            ok = t.read_file(path)   # False (not an exception) when the file's validators name another stack
        except Exception:
            ok = False
        if not ok and not tuning:    # nothing usable: TunableOp off again, the default heuristics stay
            t.enable(False)
            return False
    _state["on"] = True
    return True
