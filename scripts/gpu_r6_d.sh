#!/bin/bash
# round 6: look-back robustness (new segment order, error word, contention / CU-mask tests), designed-less ABI, whole suite
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lookback.py -x -q 2>&1 | tail -15 | tee $out/pytest_lookback.log
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $out/pytest_gpu.log
timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | tee $out/small_eq_steps.log
timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1 | tee $out/small_dyn_steps.log
