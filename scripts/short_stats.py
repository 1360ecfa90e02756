"""Prints a rocprofv3 kernel_stats csv with shortened kernel names: name, calls, total ms, average us, percent."""
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
print(f"{'kernel':70s} {'calls':>8s} {'total ms':>10s} {'avg us':>9s} {'%':>6s}")
for r in rows[1:]:
    if len(r) < 5:
        continue
    name = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", r[0])
    name = re.sub(r"<.*", "", name) if not name.startswith("Cijk") else name[:40]
    print(f"{name[:70]:70s} {int(r[1]):8d} {float(r[2]) / 1e6:10.2f} {float(r[3]) / 1e3:9.2f} {float(r[4]):6.2f}")
tot = sum(float(r[2]) for r in rows[1:] if len(r) >= 5)
print(f"total kernel time {tot / 1e6:.1f} ms over {sum(int(r[1]) for r in rows[1:] if len(r) >= 5)} launches")
