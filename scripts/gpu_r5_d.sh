#!/bin/bash
# round 5, call D: prefetched tail operands; torch.ops.dasp.* for the reference's signatures (GPU suite incl. opcheck / compile), eager vs graph
out=gpurun_out/r05d; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -60 | tee $out/pytest_gpu.log
for rep in 1 2; do for m in 0 1; do DASP_SEG_GRAM=$m timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1; done; done | tee $out/seg_gram_fused_ab.log
for shape in "8 2 131072" "16 2 131072" "32 2 131072"; do DASP_HIP_LIB=$PWD/tools/trace/libdasp_hip.so DASP_TORCH_OPS=0 timeout 200 python scripts/seg_tail_trace.py $shape 2>&1 | tail -1; done | tee $out/seg_tail_trace.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05d/bench.json").read().strip().splitlines()[-1])
print("headline ms", round(d["ms_per_step"], 4), d["launch_ms_per_step"], "bwd", d["roofline"]["ms"], "fwd", d["roofline_fwd"]["ms"])
s = d["secondary"]
for k, v in s.items():
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in ("ms_fwd_bwd", "gpu_ms_fwd_bwd", "ms_fwd_bwd_graph", "ms_fwd_bwd_wall", "host_randn_ms", "h2d_copy_ms", "eager_ms_by_binding", "ms_fwd_bwd_gpu_bound") if kk in v})
PY
