export TMPDIR=/tmp
for v in base lean; do
  rm -rf gpurun_out/clk_$v
  LIB=$PWD/brl_amd/lib/variants/$v.so timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk_$v -- python3 scripts/pmc_run.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
dur = {}
for f in glob.glob("gpurun_out/clk_$v/**/*kernel_trace*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rollout" in r["Kernel_Name"]:
            dur.setdefault(r.get("Dispatch_Id") or r.get("Correlation_Id"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
cnt = {}
for f in glob.glob("gpurun_out/clk_$v/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rollout" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[r.get("Dispatch_Id") or r.get("Correlation_Id")] = float(r["Counter_Value"])
ds = sorted(dur.values()); cs = sorted(cnt.values())
print("$v", "kernel ns", ds, "GRBM_GUI_ACTIVE", cs)
if ds and cs:
    print("$v", "effective clock GHz ~", (sum(cs)/len(cs)) / 8 / (sum(ds)/len(ds)))
PY
  rm -rf gpurun_out/clk_$v
done
