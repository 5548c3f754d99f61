#!/bin/bash
out=gpurun_out/r05n; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sosfilt.py tests/test_gpu_chain.py tests/test_gpu_graph_replay.py tests/test_gpu_torch_ops.py -q -m gpu --tb=short -x > $out/pytest.log 2>&1; grep -v "frame #" $out/pytest.log | grep -E "passed|failed|Error|FAILED|core|assert" | tail -6
for rep in 1 2; do
  DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back gathered by the workgroup, first tiles requested early/'
  DASP_HIP_LIB=$PWD/tools/lbold/libdasp_hip.so DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back by wave 0/'
done | tee $out/lookback_gather_ab.log
for lib in "" "$PWD/tools/lbold/libdasp_hip.so"; do
echo "lib: $lib"
( cd /tmp && DASP_HIP_LIB=$lib DASP_TORCH_OPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/small_eq_kernels.py > /dev/null 2> $GRAFT_REPO_ROOT/$out/rp.err )
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05n/rp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(r["Name"][:64], r["Calls"], r["AverageNs"])
PY
rm -rf $out/rp
done 2>&1 | tee -a $out/lookback_gather_ab.log
FUZZ_EQ_ONLY=1 FUZZ_SECONDS=40 timeout 300 python scripts/fuzz_gpu.py 11 2>&1 | tail -4 | cut -c1-250
