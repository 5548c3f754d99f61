#!/bin/bash
# round 6, first GPU call: the new mtrand tests, its timing, then the whole GPU suite and the bench line as a baseline for the round
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mtrand.py -x -q 2>&1 | tail -15 | tee $out/pytest_mtrand.log
timeout 300 python scripts/mtrand_time.py 8 16 128 2>&1 | tail -5 | tee $out/mtrand_time.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/mtrand_time.py 8 128 > /dev/null 2> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/mtrand_kernel_stats.csv; rm -rf $out/rprof; head -8 $out/mtrand_kernel_stats.csv
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $out/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_a.json 2> $out/bench_a.err; tail -c 1500 $out/bench_a.json
