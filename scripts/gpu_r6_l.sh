#!/bin/bash
# repeat runs on the final sources: the random stream's fuzz, the look-back and random-stream tests five times over, the all-ops fuzz
out=gpurun_out/r06; mkdir -p $out
timeout 400 python scripts/fuzz_mtrand.py 120 1 2>&1 | tail -5 | tee $out/fuzz_mtrand.log
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_lookback.py tests/test_gpu_mtrand.py -q 2>&1 | tail -1; done | tee $out/repeat_lookback_mtrand.log
timeout 400 python scripts/fuzz_gpu.py 7 120 2>&1 | tail -25 | tee $out/fuzz_all_ops.log
