#!/bin/bash
out=gpurun_out/r05m; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_graph_replay.py tests/test_gpu_dynamics.py tests/test_gpu_chain.py tests/test_gpu_modules.py tests/test_gpu_torch_ops.py -q -m gpu --tb=short > $out/pytest.log 2>&1; grep -v "frame #" $out/pytest.log | grep -E "passed|failed|Error|FAILED|core|assert" | tail -12
for rep in 1 2; do
  DASP_TORCH_OPS=0 timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1
  DASP_HIP_LIB=$PWD/tools/nodynlb/libdasp_hip.so DASP_TORCH_OPS=0 timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1
done | tee $out/dyn_lookback_ab.log
timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1 | sed 's/in-tree/in-tree, torch ops/' | tee -a $out/dyn_lookback_ab.log
