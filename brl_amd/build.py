"""Builds libbrl_hip.so (gfx950) in-tree with hipcc.  `python -m brl_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "brl_kernels.hip")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")


def deps():
    """every source the library is built from: csrc/*.hip, csrc/*.hpp, include/*.h"""
    import glob
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.hpp"))
                  + glob.glob(os.path.join(INCLUDE, "*.h")))


OUT = os.path.join(PKG, "lib", "libbrl_hip.so")

# -ffp-contract=off: GAE / reward arithmetic must round like the scalar oracle (no FMA fusion)
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        cmd = [hipcc()] + FLAGS + ["-o", OUT, SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
