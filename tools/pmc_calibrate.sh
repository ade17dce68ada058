#!/bin/bash
# Runs tools/pmc_calibrate.hip under rocprofv3 (FETCH_SIZE and WRITE_SIZE in separate passes) on the GPU box:
#   bash tools/pmc_calibrate.sh <tag>   ->  gpurun_out/<tag>_pmc_calibration.json
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/cal_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/pmc_calibrate.hip -o /tmp/pmc_calibrate || exit 1
cd /tmp
/tmp/pmc_calibrate > $R/gpurun_out/cal_$TAG/known.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/cal_$TAG/$c -o $c -- /tmp/pmc_calibrate > /dev/null 2>&1
done
python - <<PY
import csv, glob, json, os
root = "$R/gpurun_out/cal_$TAG"
known = json.load(open(os.path.join(root, "known.json")))
out = {}
for path in glob.glob(os.path.join(root, "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].strip()
        out.setdefault(k, {})[row["Counter_Name"]] = float(row["Counter_Value"]) * 1024      # KiB -> bytes
for k, v in out.items():
    kn = known.get(k, {})
    if kn.get("read"):
        v["known_read"] = kn["read"]; v["FETCH_over_known"] = round(v.get("FETCH_SIZE", 0) / kn["read"], 4)
    if kn.get("written"):
        v["known_written"] = kn["written"]; v["WRITE_over_known"] = round(v.get("WRITE_SIZE", 0) / kn["written"], 4)
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_calibration.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
