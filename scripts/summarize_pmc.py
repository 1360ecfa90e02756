"""Summarises rocprofv3 --pmc CSVs (WRITE_SIZE / FETCH_SIZE passes) per kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for tag, counter in (("prof_pmc_w", "WRITE_SIZE"), ("prof_pmc_r", "FETCH_SIZE")):
    files = glob.glob(os.path.join(out, tag, "**", "*counter_collection*.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            acc[row["Kernel_Name"]][0] += float(row["Counter_Value"])
            acc[row["Kernel_Name"]][1] += 1
    for k, (v, n) in acc.items():
        short = k.split("(")[0][:60]
        res.setdefault(short, {})[counter] = {"mean_raw": v / max(n, 1), "launches": n}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "pmc_raw.json"), "w"), indent=1)
