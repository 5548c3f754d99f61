#!/bin/bash
# Gram-matrix backward: gradient tests with it on, then the bench line with it off / on (same box)
out=gpurun_out/gram1; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_sosfilt.py tests/test_gpu_modules.py tests/test_gpu_chain.py tests/test_gpu_torch_ops.py -q -m gpu --tb=short 2>&1 | grep -E "^E  .*assert|FAILED|passed|failed" | cut -c1-300 | tee $out/pytest.log
for v in 0 1 0 1; do
  DASP_BWD_GRAM=$v timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2> $out/bench_$v.err | tee -a $out/bench_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gram=$v ms', d['ms_per_step'], 'passes', d.get('passes_ms'), 'iso', d.get('isolated_events_ms'))"
done
