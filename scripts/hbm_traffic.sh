#!/bin/bash
# HBM bytes per launch of the two EQ kernels from the PMC counters, tied to the kernel sources that were measured.
#   scripts/hbm_traffic.sh profiles/r02          (on the GPU box; writes <dir>/hbm_traffic.json)
# Two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") with
# --kernel-trace only, on tools/sosbench at the north-star shape with the designed-cascade backward kernel (what parametric_eq runs).
# FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md "HBM"); WRITE_SIZE as is.
set -e
out=${1:-profiles/r02}
mkdir -p "$out" gpurun_out/pmc_r2
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_r2/$c
  DASP_PEQ=1 DASP_SPLIT_FINALIZE=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r2/$c -o p -- \
      ./tools/sosbench 256 2 131072 40 > gpurun_out/pmc_r2/$c.log 2>&1 || true
done
python3 - "$out" <<'PY'
import csv, glob, hashlib, json, os, sys
out = sys.argv[1]
def counter(name):
    path = glob.glob(f"gpurun_out/pmc_r2/{name}/**/*counter_collection.csv", recursive=True)[0]
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            n = r["Kernel_Name"]
            k = "sos_fwd_kernel" if "sos_fwd_kernel" in n else "sos_bwd_kernel" if ("sos_bwd_gram_kernel" in n or "sos_bwd_kernel" in n) else None      # (the backward kernel of the build: Gram-matrix or recomputation)
            if k:
                vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in vals.items()}      # second half of the launches (warm)
f, w = counter("FETCH_SIZE"), counter("WRITE_SIZE")
sys.path.insert(0, os.getcwd())
from dasp_pytorch_amd.csrc.build import kernel_source_hash
units = 256 * 2 * 131072
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes with --kernel-trace only, KB units) on tools/sosbench 256 2 131072 "
               "(DASP_PEQ=1; backward = sos_bwd_gram_kernel), averaged over the second half of 42 launches; FETCH_SIZE doubled "
               "per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads), WRITE_SIZE as is. scripts/hbm_traffic.sh",
       "shape": [256, 2, 131072], "kernel_source_hash": kernel_source_hash()}
for k, alg in (("sos_fwd_kernel", 8 * units), ("sos_bwd_kernel", 12 * units)):
    res[k] = {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w[k], "hbm_bytes": int(2 * f[k] * 1024 + w[k] * 1024), "algorithmic_bytes": alg}
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(res))
PY
