#!/bin/bash
# round 2, GPU call B: the whole GPU suite after the boundary changes
mkdir -p gpurun_out/r2b
cd /root/repo
python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r2b/pytest_gpu.log
tail -40 gpurun_out/r2b/pytest_gpu.log
