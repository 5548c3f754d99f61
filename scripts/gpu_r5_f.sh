#!/bin/bash
# round 5, call F: after the clean-up (recomputation kernels, radix-3 frames and every getenv of csrc gone): GPU suite, binding check of
# sosfilt, timings
out=gpurun_out/r05f; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --tb=short > $out/pytest_gpu_full.log 2>&1; grep -v "frame #" $out/pytest_gpu_full.log | grep -E "^E  |^FAILED|passed|failed|^tests/.*Error" | head -60
timeout 300 python scripts/debug_sosfilt_bindings.py 2>&1 | grep -v Warn | tail -12 | tee $out/debug_sosfilt.log
for rep in 1 2; do timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1; done | tee $out/seg_gram_fused.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05f/bench.json").read().strip().splitlines()[-1])
print("headline ms", round(d["ms_per_step"], 4), d["launch_ms_per_step"], "bwd", d["roofline"]["ms"], "fwd", d["roofline_fwd"]["ms"], d["isolated_events_ms"])
s = d["secondary"]
for k, v in s.items():
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in ("ms_fwd_bwd", "gpu_ms_fwd_bwd", "ms_fwd_bwd_graph", "ms_fwd_bwd_wall", "eager_ms_by_binding") if kk in v})
PY
