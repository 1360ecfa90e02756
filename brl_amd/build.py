"""Builds libbrl_hip.so (gfx950) in-tree with hipcc.  `python -m brl_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
OUT = os.path.join(PKG, "lib", "libbrl_hip.so")
OBJ_DIR = os.path.join(PKG, "lib", "obj")   # one object per translation unit (git-ignored)

# -ffp-contract=off: GAE / reward arithmetic must round like the scalar oracle (no FMA fusion)
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17"]


def units():
    """the translation units of the library: every csrc/*.hip"""
    import glob
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))


def headers():
    import glob
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h")))


def deps():
    """every source the library is built from: csrc/*.hip, csrc/*.hpp, include/*.h"""
    return units() + headers()


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
    """compiles the translation units that changed (a header change rebuilds all of them: they are few) and links them"""
    if not (force or needs_build()):
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in headers())
    procs, objs = [], []
    for src in units():
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            cmd = [hipcc()] + FLAGS + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:   # (the units compile side by side)
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
