#!/bin/bash
# round 6: stability of the new pieces under repetition (look-back beside busy CUs / CU masks, mtrand), fuzz of every op on the round's sources
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_lookback.py tests/test_gpu_mtrand.py -x -q 2>&1 | tail -1; done | tee $out/repeat_lookback_mtrand.log
FUZZ_SECONDS=150 timeout 600 python scripts/fuzz_gpu.py 11 > $out/fuzz_all_ops.log 2>&1; tail -22 $out/fuzz_all_ops.log | cut -c1-200
FUZZ_EQ_ONLY=1 FUZZ_SECONDS=60 timeout 300 python scripts/fuzz_gpu.py 13 > $out/fuzz_eq.log 2>&1; tail -3 $out/fuzz_eq.log | cut -c1-300
