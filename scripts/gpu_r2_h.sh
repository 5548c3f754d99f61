#!/bin/bash
# round 2, GPU call H: band-split filter bank (tests + timing), small-batch table
mkdir -p gpurun_out/r2h
cd /root/repo
python -m pytest tests/test_gpu_reverb.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r2h/pytest.log; tail -4 gpurun_out/r2h/pytest.log
for s in 1 0; do
  if [ $s = 1 ]; then export DASP_REVERB_BAND_SPLIT=1; else unset DASP_REVERB_BAND_SPLIT; fi
  python scripts/reverb_time.py 8 2 131072 2>&1 | grep chunk >> gpurun_out/r2h/reverb_small.log
  python scripts/reverb_time.py 16 2 131072 2>&1 | grep chunk >> gpurun_out/r2h/reverb_small.log
  python scripts/reverb_time.py 128 2 262144 2>&1 | grep chunk >> gpurun_out/r2h/reverb_small.log
done
cat gpurun_out/r2h/reverb_small.log
python scripts/small_batch2.py > gpurun_out/r2h/small_batch.log 2>&1; grep -v amdgpu gpurun_out/r2h/small_batch.log
