#!/bin/bash
# round 4, A/B 1: hand-off protocol (release/acquire counter vs round-3 relaxed) and fused finalize on the full-batch path
out=gpurun_out/r4_ab1.log; : > $out
for v in in-tree relaxed fusedfin fusedfin_relaxed; do
  if [ $v = in-tree ]; then unset DASP_HIP_LIB; else export DASP_HIP_LIB=$PWD/tools/$v/libdasp_hip.so; fi
  for rep in 1 2; do python scripts/small_batch_graph.py $v >> $out 2>>gpurun_out/r4_ab1.err; done
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>>gpurun_out/r4_ab1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'tag': '$v', 'ms_per_step': d['ms_per_step'], 'launch': d['launch_ms_per_step'], 'blocks': d['block_ms_per_step'], 'bwd_ms': d['roofline']['ms'], 'fwd_ms': d['roofline_fwd']['ms'], 'small': d['small_kernels_ms']}))" >> $out
done
unset DASP_HIP_LIB
python -m pytest tests/test_gpu_sosfilt.py tests/test_gpu_dynamics.py tests/test_gpu_chain.py tests/test_gpu_reverb.py tests/test_gpu_fp64.py -m gpu -q -x 2>&1 | tail -5 >> $out
