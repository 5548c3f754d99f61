#!/bin/bash
# per-kernel GPU time of the reverb's fwd + bwd step at (B, C, N) under rocprofv3, for the frame plans DASP_REVERB_RADIX3 = 1 / 0
# usage: bash scripts/rv_kernels.sh [B C N] -> gpurun_out/rv_kernels.log
B=${1:-128}; C=${2:-2}; N=${3:-262144}
out=$GRAFT_REPO_ROOT/gpurun_out; : > $out/rv_kernels.log
for v in 1 0; do
  rm -rf /tmp/rvk; ( cd /tmp && export TMPDIR=/tmp && DASP_REVERB_RADIX3=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rvk -o p -- python $GRAFT_REPO_ROOT/scripts/reverb_run.py $B $C $N 8 > /dev/null 2>&1 )
  echo "== DASP_REVERB_RADIX3=$v ($B,$C,$N)" >> $out/rv_kernels.log
  python - $(find /tmp/rvk -name "*kernel_stats.csv" | head -1) >> $out/rv_kernels.log <<'PY'
import csv, sys, re
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dasp::" not in n: continue
    name = re.sub(r"\(.*", "", n).replace("void dasp::", "")
    us = float(r["AverageNs"]) / 1e3; calls = int(r["Calls"])
    per_step = us * calls / 8
    tot += per_step
    print(f"{name:44s} calls/step {calls / 8:4.1f}  avg {us:8.1f} us  per step {per_step:8.1f} us")
print(f"sum per step {tot:.1f} us")
PY
done
