#!/bin/bash
out=gpurun_out/r05g; mkdir -p $out
echo "== in-tree"; timeout 300 python scripts/debug_sosfilt_bindings.py 2>&1 | grep -v Warn | tail -20 | tee $out/debug_sosfilt.log

timeout 600 python -m pytest tests/test_gpu_torch_ops.py tests/test_gpu_sosfilt.py tests/test_gpu_dynamics.py -q -m gpu --tb=short 2>&1 | grep -v "frame #" | tail -6
FUZZ_SECONDS=100 timeout 400 python scripts/fuzz_gpu.py 7 2>&1 | tail -30 | tee $out/fuzz_all_ops.log
