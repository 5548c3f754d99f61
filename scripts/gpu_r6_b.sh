#!/bin/bash
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mtrand.py -x -q 2>&1 | tail -5 | tee $out/pytest_mtrand.log
timeout 300 python scripts/mtrand_time.py 8 16 128 2>&1 | tail -3 | tee $out/mtrand_time.log
for b in 8 128; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/mtrand_time.py $b > /dev/null 2> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/mtrand_kernel_stats_b$b.csv; rm -rf $out/rprof; head -4 $out/mtrand_kernel_stats_b$b.csv | cut -c1-200
done
