"""One macro-step of the policy rollout as the GPU ran it: kernel start / duration / gap after the previous kernel, from a rocprofv3
kernel trace of scripts/prof_policy_rollout.py.  usage: python scripts/rollout_timeline.py <kernel_trace.csv> [sub-step index]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_policy_step" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 40
a, b = idx[k] + 1, idx[k + 8] + 1          # eight sub-steps = two macro-steps
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
gaps = 0.0
print(f"{'kernel':40s} {'start us':>9s} {'dur us':>8s} {'gap us':>8s}")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:40]
    g = None if prev_end is None else (s - prev_end) / 1e3
    gaps += g or 0.0
    print(f"{name:40s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {'' if g is None else format(g, '8.2f'):>8s}")
    prev_end = e
print(f"eight sub-steps = {(prev_end - t0) / 1e3:.2f} us, of which gaps {gaps:.2f} us")
