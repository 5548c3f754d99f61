#!/bin/bash
# round 2, GPU call N: filter bank with the envelope inside the transform - parity, timing against the per-band route, kernel stats
mkdir -p gpurun_out/r2n; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_reverb.py tests/test_gpu_modules.py -x -q 2>&1 | tail -8 | tee gpurun_out/r2n/pytest.log
for lim in 2 0; do DASP_REVERB_WEIGHT_LIMIT=$lim timeout 300 python scripts/reverb_time.py 128 2 262144 2>&1 | tail -1; done | tee gpurun_out/r2n/reverb_time.log
for lim in 2 0; do DASP_REVERB_WEIGHT_LIMIT=$lim timeout 300 python scripts/reverb_time.py 8 2 131072 2>&1 | tail -1; done | tee -a gpurun_out/r2n/reverb_time.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2n/prof -o p -- python scripts/reverb_time.py 128 2 262144 > /dev/null 2> gpurun_out/r2n/prof.err
cp $(find gpurun_out/r2n/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r2n/reverb_kernel_stats.csv; rm -rf gpurun_out/r2n/prof
head -16 gpurun_out/r2n/reverb_kernel_stats.csv | cut -c1-150
