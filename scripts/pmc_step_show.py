"""Per-kernel means of the counters scripts/pmc_step.sh collected (the last 15 x 8 dispatches = the timed steps dominate the mean)."""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection*.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")[:56]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
cols = sorted({c for d in acc.values() for c in d})
print(f"{'kernel':56s} {'n':>6s} " + " ".join(f"{c[:20]:>20s}" for c in cols))
for k, d in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = max(len(v) for v in d.values())
    if n < 8:
        continue
    print(f"{k:56s} {n:6d} " + " ".join((f"{sum(d[c]) / len(d[c]):20.1f}" if c in d else f"{'-':>20s}") for c in cols))
