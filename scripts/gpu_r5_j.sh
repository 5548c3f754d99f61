#!/bin/bash
# round 5, call J: single-launch segmented backward (look-back inside the Gram kernel) against pre-pass launch + pass
out=gpurun_out/r05j; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sosfilt.py tests/test_gpu_chain.py tests/test_gpu_modules.py tests/test_gpu_torch_ops.py -q -m gpu --tb=short -x 2>&1 | grep -v "frame #" | tail -8
for rep in 1 2; do
  DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back both directions/'
  DASP_HIP_LIB=$PWD/tools/nobwdlookback/libdasp_hip.so DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/default/look-back forward only/'
done | tee $out/bwd_lookback_ab.log
FUZZ_SECONDS=60 FUZZ_EQ_ONLY=1 timeout 300 python scripts/fuzz_gpu.py 14 2>&1 | tail -4
