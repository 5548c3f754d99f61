#!/bin/bash
# round 5, call E: failing torch-op tests with full tracebacks; coalesced basis layout timings
out=gpurun_out/r05e; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_torch_ops.py -q -m gpu --tb=long -k "reference_signatures or functional_calls or range_flag" > $out/pytest_torch_ops_full.log 2>&1
grep -v "frame #" $out/pytest_torch_ops_full.log | grep -E "^E |Error|error|FAILED|passed|failed|^tests/|^/root|dasp_torch_ops|ops.py|functional.py" | head -150
timeout 600 python -m pytest tests/test_gpu_sosfilt.py tests/test_gpu_modules.py tests/test_gpu_chain.py -q -m gpu --tb=short 2>&1 | tail -5
for rep in 1 2; do for m in 0 1; do DASP_SEG_GRAM=$m timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1; done; done | tee $out/seg_gram_fused_ab.log
for shape in "8 2 131072" "16 2 131072" "32 2 131072"; do DASP_HIP_LIB=$PWD/tools/trace/libdasp_hip.so DASP_TORCH_OPS=0 timeout 200 python scripts/seg_tail_trace.py $shape 2>&1 | tail -1; done | tee $out/seg_tail_trace.log
