#!/bin/bash
# round 2, final GPU call: full GPU test suite, the profile set for profiles/r02 (scripts/gpu_r2_g.sh), the secondary logs
mkdir -p gpurun_out/r02; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r02/pytest_gpu.log
bash scripts/gpu_r2_g.sh 2>&1 | tail -30
timeout 300 python scripts/sections_sweep.py > gpurun_out/r02/sections_per_call.log 2>&1
timeout 300 python scripts/small_batch2.py > gpurun_out/r02/small_batch_gpu_times.log 2>&1
for s in "128 2 262144" "32 2 131072" "16 2 131072" "8 2 131072"; do timeout 300 python scripts/reverb_time.py $s 2>&1 | tail -1; done > gpurun_out/r02/reverb_times.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/rprof -o p -- python scripts/reverb_time.py 128 2 262144 > /dev/null 2> gpurun_out/r02/rprof.err
cp $(find gpurun_out/r02/rprof -name "*kernel_stats.csv" | head -1) gpurun_out/r02/reverb_kernel_stats.csv; rm -rf gpurun_out/r02/rprof
FUZZ_SECONDS=150 timeout 400 python scripts/fuzz_gpu.py 21 > gpurun_out/r02/fuzz_all.log 2>&1
FUZZ_EQ_ONLY=1 FUZZ_SECONDS=120 timeout 400 python scripts/fuzz_gpu.py 22 > gpurun_out/r02/fuzz_eq.log 2>&1
tail -3 gpurun_out/r02/sections_per_call.log; tail -4 gpurun_out/r02/small_batch_gpu_times.log; cat gpurun_out/r02/reverb_times.log; grep -c FAIL gpurun_out/r02/fuzz_all.log; tail -4 gpurun_out/r02/fuzz_eq.log
