"""Every kernel of a rocprofv3 kernel trace in start order with its duration and the idle time in front of it; a summary of
where the GPU sat idle.  usage: python scripts/trace_gaps.py <kernel_trace.csv> [first] [count]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
prev_end = None
busy = idle = 0.0
big = []
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
    gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
    if first <= i < first + count:
        print(f"{i:6d} {name:60s} {(e - s) / 1e3:9.2f} {gap:9.2f}")
    busy += (e - s) / 1e3
    if prev_end is not None and gap < 2000:
        idle += gap
        if gap > 15:
            big.append((gap, i, name))
    prev_end = e if prev_end is None else max(prev_end, e)
print(f"kernels {len(rows)}, busy {busy / 1e3:.2f} ms, idle between kernels (gaps < 2 ms) {idle / 1e3:.2f} ms")
print("largest gaps:", sorted(big, reverse=True)[:12])
