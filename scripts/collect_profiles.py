"""Copies the judged summaries of a scripts/gpu_round.sh run from gpurun_out/<tag> into profiles/<round>/
and refreshes profiles/pmc_traffic.json (read by bench.py for roofline.traffic)."""
import json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
for f in ("kernel_stats.csv", "pmc_raw.json", "pmc_mix.txt", "bench.json", "pytest_gpu.txt", "smoke.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
pmc = json.load(open(os.path.join(src, "pmc_raw.json")))
roll = next(v for k, v in pmc.items() if "rollout" in k)
w_kb = roll["WRITE_SIZE"]["mean_raw"]
r_kb = roll["FETCH_SIZE"]["mean_raw"]
out = {
    "source": f"rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes, {tag} (scripts/gpu_round.sh)",
    "kernel": next(k for k in pmc if "rollout" in k),
    "WRITE_SIZE_KB_per_launch": w_kb,
    "FETCH_SIZE_KB_per_launch_raw": r_kb,
    "note": "WRITE_SIZE is exact for 16-B-per-lane streaming stores; FETCH_SIZE is doubled (gfx950 reports 1/2 of "
            "coalesced reads; MI355X_MICROARCH.md HBM section) — the kernel's reads are sparse 16-B LUT rows + the "
            "1 MB table state, so the doubled figure is an upper estimate",
    "rollout_hbm_bytes_per_launch": int(w_kb * 1024 + 2 * r_kb * 1024),
    "rollout_write_bytes_per_launch": int(w_kb * 1024),
}
json.dump(out, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
