#!/bin/bash
# round 5, call A: the segmented Gram pass with its finalize inside the launch (basis workgroups + last-workgroup tail) against round 4's
# kernels (DASP_SEG_GRAM=0 while both exist): GPU suite, graph-step timings, fuzz sweeps on library defaults, per-kernel durations at (16,2,131072).
out=gpurun_out/r05a; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $out/pytest_gpu.log
for rep in 1 2; do for m in 0 1; do DASP_SEG_GRAM=$m timeout 300 python scripts/seg_gram_ab.py 2>/dev/null | tail -1; done; done | tee $out/seg_gram_fused_ab.log
FUZZ_SECONDS=100 FUZZ_EQ_ONLY=1 timeout 400 python scripts/fuzz_gpu.py 11 2>&1 | tail -12 | tee $out/fuzz_eq.log
FUZZ_SECONDS=140 timeout 500 python scripts/fuzz_gpu.py 5 2>&1 | tail -25 | tee $out/fuzz_all_ops.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/small_eq_kernels.py > $GRAFT_REPO_ROOT/$out/small_eq.out 2> $GRAFT_REPO_ROOT/$out/rp.err )
cp $(find $out/rp -name "*kernel_stats.csv" | head -1) $out/small_eq_kernel_stats.csv; rm -rf $out/rp
cut -d, -f1-4 $out/small_eq_kernel_stats.csv | head -12
