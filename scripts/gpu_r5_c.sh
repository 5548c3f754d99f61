#!/bin/bash
# round 5, call C: lag sums per workgroup in the fused tail, write-through gx stores in the segmented Gram pass
out=gpurun_out/r05c; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | grep -v Warning | tail -30 | tee $out/pytest_gpu.log
for rep in 1 2; do for m in 0 1; do DASP_SEG_GRAM=$m timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1; done; done | tee $out/seg_gram_fused_ab.log
for shape in "8 2 131072" "16 2 131072" "32 2 131072"; do DASP_HIP_LIB=$PWD/tools/trace/libdasp_hip.so DASP_TORCH_OPS=0 timeout 200 python scripts/seg_tail_trace.py $shape 2>&1 | tail -1; done | tee $out/seg_tail_trace.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/small_eq_kernels.py > $GRAFT_REPO_ROOT/$out/small_eq.out 2> $GRAFT_REPO_ROOT/$out/rp.err )
cp $(find $out/rp -name "*kernel_stats.csv" | head -1) $out/small_eq_kernel_stats.csv; rm -rf $out/rp
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r05c/small_eq_kernel_stats.csv")))[:6]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
FUZZ_SECONDS=60 FUZZ_EQ_ONLY=1 timeout 300 python scripts/fuzz_gpu.py 12 2>&1 | tail -5 | tee $out/fuzz_eq.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $out/bench_headline.json 2> $out/bench_headline.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05c/bench_headline.json").read().strip().splitlines()[-1])
print("headline ms", round(d["ms_per_step"], 4), d["launch_ms_per_step"], "bwd", d["roofline"]["ms"], "fwd", d["roofline_fwd"]["ms"], d["isolated_events_ms"])
PY
