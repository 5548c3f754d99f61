#!/bin/bash
# round 5 GPU call: full GPU test suite, smoke, the bench line at the driver's arguments and at the defaults, rocprofv3 kernel stats of the
# bench command, reverb kernel stats, the small-batch tables, fuzz. usage (gpurun): bash scripts/gpu_r5_final.sh [skip-tests]
out=gpurun_out/r05; mkdir -p $out; export TMPDIR=/tmp
if [ "$1" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee $out/smoke.log
fi
# (the PMC passes run in a call of their own, scripts/gpu_r5_pmc.sh: their hbm_traffic*.json are committed under profiles/r05 before this call)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv; rm -rf $out/rprof
( cd /tmp && DASP_RV_NOISE=generated rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/reverb_time.py 128 2 262144 > /dev/null 2>> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/reverb_kernel_stats.csv; rm -rf $out/rprof
( cd /tmp && DASP_TORCH_OPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/small_eq_kernels.py > /dev/null 2>> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/small_eq_kernel_stats.csv; rm -rf $out/rprof
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/chain_kernel_count.py > /dev/null 2>> $GRAFT_REPO_ROOT/$out/rprof.err )
python scripts/kernel_count_report.py $out/rprof 50 > $out/chain_kernel_count.log; rm -rf $out/rprof
DASP_TORCH_OPS=0 timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 | sed 's/"DASP_SEG_GRAM": "default"/"what": "parametric_eq fwd+bwd, graph step ms"/' > $out/small_batch_steps.log
timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1 | sed 's/"lib": "in-tree"/"what": "compressor fwd+bwd, graph step ms"/' >> $out/small_batch_steps.log
if [ "$1" != "skip-tests" ]; then
  FUZZ_SECONDS=150 timeout 600 python scripts/fuzz_gpu.py 7 > $out/fuzz_all_ops.log 2>&1; tail -25 $out/fuzz_all_ops.log | cut -c1-200
  FUZZ_EQ_ONLY=1 FUZZ_SECONDS=60 timeout 300 python scripts/fuzz_gpu.py 9 > $out/fuzz_eq.log 2>&1; tail -3 $out/fuzz_eq.log | cut -c1-300
fi
python - <<'PY'
import json
for f in ("bench_driver_args", "bench", "bench_under_rocprof"):
    try:
        d = json.loads(open(f"gpurun_out/r05/{f}.json").read().strip().splitlines()[-1])
        print(f, "ms", round(d["ms_per_step"], 4), "value %.4g" % d["value"], d["launch_ms_per_step"], "bwd", d["roofline"]["ms"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"],
              "fwd", d["roofline_fwd"]["ms"], "both", d["roofline_fwd_bwd"]["frac"])
        for k, v in (d.get("secondary") or {}).items():
            print("  ", k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "wall_ms_per_step", "eager_wall_ms", "graph_ms", "ms", "value")} if isinstance(v, dict) else v)
    except Exception as e:
        print(f, "ERR", e)
PY
cat $out/small_batch_steps.log
