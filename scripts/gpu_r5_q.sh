#!/bin/bash
out=gpurun_out/r05q; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dynamics.py tests/test_gpu_torch_ops.py tests/test_gpu_graph_replay.py tests/test_gpu_chain.py tests/test_gpu_modules.py -q -m gpu --tb=short > $out/pytest.log 2>&1; grep -v "frame #" $out/pytest.log | grep -E "passed|failed|Error|FAILED|core|assert" | tail -8
for rep in 1 2; do
  timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1 | sed 's/in-tree/six control vectors straight into the kernels (torch ops)/'
  DASP_TORCH_OPS=0 timeout 300 python scripts/dyn_small_ab.py 2>/dev/null | tail -1 | sed 's/in-tree/stacked (B, 5) rows (ctypes binding)/'
done | tee $out/dyn_rows_ab.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp -o p -- python $GRAFT_REPO_ROOT/scripts/small_dyn_kernels.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python scripts/kernel_count_report.py $out/rp 60 | tee -a $out/dyn_rows_ab.log; rm -rf $out/rp
python scripts/host_cprofile.py comp 2>&1 | grep -E "wall|run_backward|dasp.dynamics" | head -4
