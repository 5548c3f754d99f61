#!/bin/bash
# HBM bytes per launch of the reverb kernels at (128, 2, 262144) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only),
# DASP_RV_NOISE selects the noise mode of scripts/reverb_time.py (default: generated inside the kernels). FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads). usage: scripts/reverb_traffic.sh <out.json>
out=${1:-gpurun_out/r03/hbm_traffic_secondary.json}
mkdir -p gpurun_out/pmc_rv "$(dirname "$out")"; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_rv/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_rv/$c -o p -- python scripts/reverb_time.py 128 2 262144 > gpurun_out/pmc_rv/$c.log 2>&1 || echo "pass $c failed"
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
def counter(name):
    path = glob.glob(f"gpurun_out/pmc_rv/{name}/**/*counter_collection.csv", recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "dasp::" in r["Kernel_Name"]:
            vals[r["Kernel_Name"].split("dasp::")[1].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in vals.items()}
f, w = counter("FETCH_SIZE"), counter("WRITE_SIZE")
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units), scripts/reverb_time.py 128 2 262144, second half of the launches; "
               "hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md). scripts/reverb_traffic.sh",
       "shape": [128, 2, 262144], "noise_mode": __import__("os").environ.get("DASP_RV_NOISE", "generated"), "kernels": {}}
tot = 0
for k in sorted(set(f) | set(w)):
    b = int(2 * f.get(k, 0) * 1024 + w.get(k, 0) * 1024)
    res["kernels"][k] = {"FETCH_SIZE_KB": f.get(k), "WRITE_SIZE_KB": w.get(k), "hbm_bytes": b}
    tot += b
res["hbm_bytes_per_step"] = tot
res["compulsory_bytes_per_step"] = int(2 * 1.354e9)
res["ratio"] = round(tot / (2 * 1.354e9), 2)
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes"] / 1e6) for k, v in res["kernels"].items()}), "total MB", round(tot / 1e6), "ratio", res["ratio"])
PY
rm -rf gpurun_out/pmc_rv
