#!/bin/bash
# round 6: mtrand v2 (k-way jumps, deferred read-back), config refactor (whole suite), expander at config 3, PMC traffic of the dynamics /
# elementwise kernels, bench with the median-timed secondaries
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mtrand.py tests/test_gpu_dynamics.py -x -q 2>&1 | tail -8 | tee $out/pytest_mtrand_dyn.log
timeout 300 python scripts/mtrand_time.py 8 16 128 2>&1 | tail -3 | tee $out/mtrand_time.log
for b in 8 128; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/mtrand_time.py $b > /dev/null 2> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/mtrand_kernel_stats_b$b.csv; rm -rf $out/rprof; head -4 $out/mtrand_kernel_stats_b$b.csv | cut -c1-200
done
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $out/pytest_gpu.log
timeout 900 bash scripts/ops_traffic.sh $out/hbm_traffic_ops.json 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c.json 2> $out/bench_c.err; tail -3 $out/bench_c.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_c.json").read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"], 4), d["roofline"])
for k, v in d["secondary"].items():
    if isinstance(v, dict):
        print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_fwd_bwd", "gpu_ms_fwd_bwd", "ms_fwd_bwd_graph", "ms_fwd_bwd_wall", "noise_stream_gpu_ms", "host_draw")}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"))
PY
