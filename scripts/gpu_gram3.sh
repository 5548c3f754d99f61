#!/bin/bash
# Gram-matrix backward: accuracy at the corners of the EQ's ranges (both kernel generations), randomized sweep, the sosfilt tests
out=gpurun_out/gram3; mkdir -p $out
for g in 1 0; do echo "== DASP_BWD_GRAM=$g (one workgroup per row forced)"; DASP_SOS_SEGMENT=0 DASP_BWD_GRAM=$g python scripts/eq_accuracy.py 2>&1 | tail -8; done | tee $out/eq_accuracy.log
FUZZ_EQ_ONLY=1 FUZZ_SECONDS=150 DASP_SOS_SEGMENT=0 python scripts/fuzz_gpu.py 7 2>&1 | tail -15 | tee $out/fuzz_eq.log
timeout 900 python -m pytest tests/test_gpu_sosfilt.py -q -m gpu --tb=short 2>&1 | grep -E "^E  .*assert|FAILED|passed|failed" | cut -c1-300 | tee $out/pytest.log
