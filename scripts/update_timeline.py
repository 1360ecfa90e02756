"""One PPO minibatch step as the GPU ran it: kernel start / duration / gap after the previous kernel, from a rocprofv3 kernel trace.
usage: python scripts/update_timeline.py <kernel_trace.csv> [step index] [name of the step's last kernel: k_adam_apply | k_shard_apply]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with k_adam_apply (whose extra workgroups gather the next minibatch): from the kernel behind one to the kernel behind the next
last = sys.argv[3] if len(sys.argv) > 3 else "k_adam_apply"
idx = [i for i, r in enumerate(rows) if last in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) >= 0 else len(idx) // 2
a, b = idx[k] + 1, idx[k + 1] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
print(f"{'kernel':58s} {'start us':>9s} {'dur us':>8s} {'gap us':>8s}")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:58]
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.2f}"
    print(f"{name:58s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:>8s}")
    prev_end = e if prev_end is None else max(prev_end, e)
print(f"step = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.2f} us")
