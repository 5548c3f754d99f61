#!/bin/bash
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mtrand.py -x -q 2>&1 | tail -2
for b in 8 128; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/mtrand_time.py $b > /dev/null 2> $GRAFT_REPO_ROOT/$out/rprof.err )
head -3 $(find $out/rprof -name "*kernel_stats.csv" | head -1) | cut -c1-180; rm -rf $out/rprof
done
timeout 300 python scripts/mtrand_time.py 8 128 2>/dev/null | tail -2 | cut -c1-200
