#!/bin/bash
# round 2, GPU call F: fp64 path tests, well-conditioned MR-STFT gradient, whole GPU suite, reverb kernel profile
mkdir -p gpurun_out/r2f
cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fp64.py tests/test_gpu_losses.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r2f/pytest_new.log; tail -25 gpurun_out/r2f/pytest_new.log
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r2f/pytest_all.log; tail -6 gpurun_out/r2f/pytest_all.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof_rev -o p -- python scripts/reverb_time.py > gpurun_out/r2f/rev_prof.log 2>&1
f=$(find gpurun_out/r2f/prof_rev -name "*kernel_stats.csv" | head -1); cut -c1-150 "$f" | head -16
