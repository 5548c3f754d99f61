#!/bin/bash
# round 4: the bench line at the driver's arguments and at the defaults, and the same command under rocprofv3 (kernel stats)
out=gpurun_out/r04; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv; rm -rf $out/rprof
