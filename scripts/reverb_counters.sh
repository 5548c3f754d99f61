#!/bin/bash
# Developer helper: per-kernel PMC counters of the reverb kernels at (128, 2, 262144), one rocprofv3 --pmc pass per counter group
# (with --kernel-trace only). usage (GPU box): scripts/reverb_counters.sh <outdir> [env assignments for the run]
out=${1:-gpurun_out/rvc}; shift
mkdir -p $out; export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '+')
  rm -rf $out/$tag
  env "$@" rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$tag -o p -- python scripts/reverb_time.py 128 2 262144 > $out/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "dasp::" not in n: continue
        n = n.split("dasp::")[1].split("(")[0]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v[len(v)//2:]) / len(v[len(v)//2:]), 1) for c, v in sorted(acc[k].items())})
PY
find $out -name "*.csv" -size +1M -delete; rm -rf $out/*/pass* 2>/dev/null
