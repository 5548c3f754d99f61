#!/bin/bash
# round 2, GPU call E: reverb (no saved wet, chunked passes) tests + chunk sweep; compressor tests; EQ kernels after the LDS reorder
mkdir -p gpurun_out/r2e
cd /root/repo
python -m pytest tests/test_gpu_reverb.py tests/test_gpu_dynamics.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r2e/pytest.log; tail -4 gpurun_out/r2e/pytest.log
for c in 0 16 32 64 128; do DASP_REVERB_CHUNK=$c python scripts/reverb_time.py >> gpurun_out/r2e/reverb.log 2>&1; done
python scripts/reverb_time.py >> gpurun_out/r2e/reverb.log 2>&1
python scripts/reverb_time.py 8 2 131072 >> gpurun_out/r2e/reverb.log 2>&1
DASP_REVERB_CHUNK=0 python scripts/reverb_time.py 8 2 131072 >> gpurun_out/r2e/reverb.log 2>&1
grep -v amdgpu.ids gpurun_out/r2e/reverb.log
for v in "DASP_DESIGNED=1" "DASP_NOGC=1"; do
  echo "== $v" >> gpurun_out/r2e/sosbench.log
  env DASP_PEQ=1 $v ./tools/sosbench 256 2 131072 400 >> gpurun_out/r2e/sosbench.log 2>&1
done
cat gpurun_out/r2e/sosbench.log
