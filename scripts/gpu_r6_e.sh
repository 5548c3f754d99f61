#!/bin/bash
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python scripts/fused_finalize_ab.py 2>&1 | tail -3 | tee $out/fused_finalize_ab.log
timeout 300 python scripts/fused_finalize_ab.py 128 2 131072 2>&1 | tail -1 | tee -a $out/fused_finalize_ab.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rprof -o p -- python $GRAFT_REPO_ROOT/scripts/fused_finalize_ab.py > /dev/null 2> $GRAFT_REPO_ROOT/$out/rprof.err )
cp $(find $out/rprof -name "*kernel_stats.csv" | head -1) $out/fused_finalize_kernel_stats.csv; rm -rf $out/rprof; head -8 $out/fused_finalize_kernel_stats.csv | cut -c1-250
