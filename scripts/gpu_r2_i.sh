#!/bin/bash
# round 2, GPU call I: fused normalised-parameter paths - tests + host overhead
mkdir -p gpurun_out/r2i
cd /root/repo
python -m pytest tests/test_gpu_modules.py tests/test_gpu_dynamics.py tests/test_gpu_fp64.py -x -q -m gpu 2>&1 | grep -v Warning | tail -30 > gpurun_out/r2i/pytest.log; grep -E "Error|error|passed|failed" gpurun_out/r2i/pytest.log | head -12
python scripts/host_overhead.py 16 2 131072 2>&1 | grep process_normalized > gpurun_out/r2i/host_overhead.log
python scripts/host_overhead.py 256 2 131072 2>&1 | grep process_normalized >> gpurun_out/r2i/host_overhead.log
cat gpurun_out/r2i/host_overhead.log
