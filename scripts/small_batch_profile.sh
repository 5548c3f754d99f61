export TMPDIR=/tmp; out=gpurun_out/sb; mkdir -p $out
timeout 300 python scripts/small_batch3.py 2>&1 | grep -v Warn | tail -6 > $out/small_batch_launch_diet.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/small_batch3.py > /dev/null 2> $GRAFT_REPO_ROOT/$out/rp.err )
cp $(find $out/rp -name "*kernel_stats.csv" | head -1) $out/small_batch_kernel_stats.csv; rm -rf $out/rp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/chain_kernel_count.py > $GRAFT_REPO_ROOT/$out/ck.out 2> $GRAFT_REPO_ROOT/$out/rp.err )
python scripts/kernel_count_report.py $out/rp 50 > $out/chain_kernel_count.log 2>&1; rm -rf $out/rp
cat $out/small_batch_launch_diet.log; tail -5 $out/chain_kernel_count.log
