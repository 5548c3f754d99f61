#!/bin/bash
# SQ counters of the reverb kernels at (128, 2, 262144): where their cycles go (VALU issue, LDS, waiting). Separate --pmc passes with
# --kernel-trace only. usage: bash scripts/rv_pmc.sh [tag] -> gpurun_out/rv_pmc_<tag>.log
tag=${1:-x}; out=$GRAFT_REPO_ROOT/gpurun_out/rv_pmc_$tag.log; : > $out
export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  rm -rf /tmp/rvp; ( cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/rvp -o p -- python $GRAFT_REPO_ROOT/scripts/reverb_run.py 128 2 262144 3 > /tmp/rvp.log 2>&1 ) || echo "pass failed: $grp" >> $out
  python3 - $(find /tmp/rvp -name "*counter_collection.csv" | head -1) >> $out <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        n = r["Kernel_Name"]
        if "dasp::" not in n: continue
        k = re.sub(r"\(.*", "", n).replace("void dasp::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        if not re.search("fb_fused_kernel<., 1|conv_rows_kernel<[01]|conv_cols_kernel<0|conv_load_kernel<0", k): continue
        print(k[:40].ljust(40), "  ".join(f"{c}={sum(v[len(v)//2:])/len(v[len(v)//2:]):.4g}" for c, v in acc[k].items()))
except Exception as e:
    print("no counters:", e)
PY
done
