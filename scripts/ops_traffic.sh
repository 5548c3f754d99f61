#!/bin/bash
# HBM bytes per fwd+bwd step of the compressor / expander scans and of gain / distortion from the PMC counters, tied to the kernel
# sources that were measured (north_star: "rocprof must report achieved HBM GB/s for the biquad/compressor scans"; the EQ's file is
# scripts/hbm_traffic.sh, the reverb's scripts/reverb_traffic.sh).     usage (GPU box): scripts/ops_traffic.sh <out.json>
# Two separate rocprofv3 --pmc passes per op (FETCH_SIZE and WRITE_SIZE, --kernel-trace only: MI355X_MICROARCH.md "rocprofv3 PMC slots");
# FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md "HBM"), WRITE_SIZE as is; KB units.
out=${1:-gpurun_out/r06/hbm_traffic_ops.json}
mkdir -p gpurun_out/pmc_ops "$(dirname "$out")"; export TMPDIR=/tmp
for spec in "compressor 256 2 262144" "expander 256 2 262144" "gain 256 2 131072" "distortion 256 2 131072"; do
  set -- $spec
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_ops/$1_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_ops/$1_$c -o p -- python scripts/op_steps.py $1 $2 $3 $4 12 > gpurun_out/pmc_ops/$1_$c.log 2>&1 || echo "pass $1 $c failed"
  done
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
from dasp_pytorch_amd.csrc.build import kernel_source_hash
def counter(op, name):
    path = glob.glob(f"gpurun_out/pmc_ops/{op}_{name}/**/*counter_collection.csv", recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "dasp::" in r["Kernel_Name"]:
            vals[r["Kernel_Name"].split("dasp::")[1].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v[len(v) // 2:]) / len(v[len(v) // 2:]), len(v)) for k, v in vals.items()}     # second half of the launches (warm)
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes with --kernel-trace only, KB units) on scripts/op_steps.py <op> 12 steps, "
               "second half of each kernel's launches; hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md). "
               "scripts/ops_traffic.sh",
       "kernel_source_hash": {"dynamics": kernel_source_hash(("dynamics.hip", "dyn_common.hpp", "common.hpp")),
                              "elementwise": kernel_source_hash(("elementwise.hip", "common.hpp"))}, "ops": {}}
for op, shape in (("compressor", (256, 2, 262144)), ("expander", (256, 2, 262144)), ("gain", (256, 2, 131072)), ("distortion", (256, 2, 131072))):
    try:
        f, w = counter(op, "FETCH_SIZE"), counter(op, "WRITE_SIZE")
    except Exception as e:
        res["ops"][op] = {"error": repr(e)}
        continue
    units = shape[0] * shape[1] * shape[2]
    ks, tot = {}, 0
    for k in sorted(set(f) | set(w)):
        b = int(2 * f.get(k, (0, 0))[0] * 1024 + w.get(k, (0, 0))[0] * 1024)
        ks[k] = {"FETCH_SIZE_KB": f.get(k, (None,))[0], "WRITE_SIZE_KB": w.get(k, (None,))[0], "hbm_bytes": b, "launches_per_step": f.get(k, w.get(k))[1] / 12}
        tot += int(b * ks[k]["launches_per_step"])
    res["ops"][op] = {"shape": list(shape), "kernels": ks, "hbm_bytes_per_step": tot, "algorithmic_bytes_per_step": 20 * units, "ratio": round(tot / (20 * units), 3)}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps({op: (v.get("hbm_bytes_per_step"), v.get("ratio")) for op, v in res["ops"].items()}))
PY
rm -rf gpurun_out/pmc_ops
