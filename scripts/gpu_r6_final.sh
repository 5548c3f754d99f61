#!/bin/bash
# round 6 final GPU call: full GPU suite, smoke, the bench line at the driver's arguments and at the defaults, rocprofv3 kernel stats of the
# bench command, of the device random stream and of the default-noise reverb step, the small-batch tables.  usage: bash scripts/gpu_r6_final.sh
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -9 | tee $out/smoke.log
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench.py --gpus 1 --steps 20 --warmup 5: $SECONDS s of wall time (process start to exit)" | tee $out/bench_wall.log
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv; rm -rf $out/rprof
for b in 8 128; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/mtrand_time.py $b > /dev/null 2>> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/mtrand_kernel_stats_b$b.csv; rm -rf $out/rprof
done
timeout 300 python scripts/mtrand_time.py 8 16 32 64 128 2>/dev/null | tail -5 > $out/mtrand_time.log
BS="8 16 128" bash scripts/mtrand_trace.sh > $out/mtrand_timeline.log 2>&1
[ -x tools/ubench6 ] && timeout 60 tools/ubench6 > $out/ubench6.log 2>&1
( cd /tmp && DASP_RV_NOISE=generated rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/reverb_time.py 128 2 262144 > /dev/null 2>> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/reverb_kernel_stats.csv; rm -rf $out/rprof
timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1 > $out/small_batch_steps.log
timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1 | sed 's/"lib": "in-tree"/"what": "compressor fwd+bwd, graph step ms"/' >> $out/small_batch_steps.log
python - <<'PY'
import json
for f in ("bench_driver_args", "bench", "bench_under_rocprof"):
    try:
        d = json.loads(open(f"gpurun_out/r06/{f}.json").read().strip().splitlines()[-1])
        print(f, "ms", round(d["ms_per_step"], 4), "value %.4g" % d["value"], "bwd", d["roofline"]["ms"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"],
              "fwd", d["roofline_fwd"]["ms"], "both", d["roofline_fwd_bwd"]["frac"])
        for k, v in (d.get("secondary") or {}).items():
            if isinstance(v, dict):
                print("  ", k, {kk: vv for kk, vv in v.items() if kk in ("ms_fwd_bwd", "gpu_ms_fwd_bwd", "ms_fwd_bwd_graph", "ms_fwd_bwd_wall", "noise_stream_gpu_ms")},
                      (v.get("roofline") or {}).get("frac"), ((v.get("roofline") or {}).get("traffic") or {}))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $out/small_batch_steps.log; cat $out/mtrand_time.log | cut -c1-200; head -6 $out/bench_kernel_stats.csv | cut -c1-200
