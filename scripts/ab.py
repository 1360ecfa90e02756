"""Same-box A/B of rollout-kernel builds: alternates libraries in subprocesses, reports medians."""
import json, os, subprocess, sys
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rep in range(4):
    for l in libs:
        env = dict(os.environ, LIB=os.path.abspath(l), CFGS="32x11")
        out = subprocess.run([sys.executable, "scripts/ablate3.py"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[l].append(json.loads(out[out.index("{"):])["32x11"])
for l, v in res.items():
    print(os.path.basename(l), sorted(v))
