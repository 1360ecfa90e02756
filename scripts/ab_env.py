"""Same-box A/B of rollout-kernel settings given as env specs ("BRL_ROLLOUT_PIPE=2,BRL_DEBUG=0" ...):
alternates them in subprocesses (scripts/ablate3.py), reports the sorted medians."""
import json, os, subprocess, sys
specs = sys.argv[1:]
res = {s: [] for s in specs}
for rep in range(4):
    for s in specs:
        env = dict(os.environ, CFGS="32x11")
        for kv in s.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                env[k] = v
        try:
            out = subprocess.run([sys.executable, "scripts/ablate3.py"], env=env, capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1]
            res[s].append(json.loads(out[out.index("{"):])[env["CFGS"]])
        except Exception as e:
            res[s].append(repr(e)[:80])
for s, v in res.items():
    print(s, v)
