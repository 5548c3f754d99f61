#!/bin/bash
out=gpurun_out/r05k; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sosfilt.py tests/test_gpu_chain.py tests/test_gpu_modules.py tests/test_gpu_torch_ops.py -q -m gpu --tb=short -x > $out/pytest.log 2>&1; grep -v "frame #" $out/pytest.log | grep -E "passed|failed|Error|FAILED|core" | tail -5
for rep in 1 2; do
  DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back both directions/'
  DASP_HIP_LIB=$PWD/tools/lbnat/libdasp_hip.so DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back both directions, natural order/'
  DASP_HIP_LIB=$PWD/tools/nobwdlookback/libdasp_hip.so DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back forward only/'
done | tee $out/bwd_lookback_ab.log
for lib in "" "$PWD/tools/lbnat/libdasp_hip.so" "$PWD/tools/nobwdlookback/libdasp_hip.so"; do
echo "lib: $lib"
( cd /tmp && DASP_HIP_LIB=$lib DASP_TORCH_OPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/small_eq_kernels.py > /dev/null 2> $GRAFT_REPO_ROOT/$out/rp.err )
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05k/rp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r["Name"][:64], r["Calls"], r["AverageNs"])
PY
rm -rf $out/rp
done 2>&1 | tee $out/bwd_lookback_kernels.log
