#!/bin/bash
# round 5, call H: single-launch segmented forward (look-back) against the two-launch path
out=gpurun_out/r05h; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sosfilt.py tests/test_gpu_chain.py tests/test_gpu_modules.py tests/test_gpu_torch_ops.py -q -m gpu --tb=short -x 2>&1 | grep -v "frame #" | tail -8
for rep in 1 2; do
  timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1
  DASP_HIP_LIB=$PWD/tools/nolookback/libdasp_hip.so DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/two-launch forward (ctypes)/'
  DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back forward (ctypes)/'
done | tee $out/fwd_lookback_ab.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/small_eq_kernels.py > /dev/null 2> $GRAFT_REPO_ROOT/$out/rp.err )
cp $(find $out/rp -name "*kernel_stats.csv" | head -1) $out/small_eq_kernel_stats.csv; rm -rf $out/rp
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r05h/small_eq_kernel_stats.csv")))[:6]:
    print(r["Name"][:64], r["Calls"], r["AverageNs"])
PY
FUZZ_SECONDS=60 FUZZ_EQ_ONLY=1 timeout 300 python scripts/fuzz_gpu.py 13 2>&1 | tail -4
