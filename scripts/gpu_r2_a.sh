#!/bin/bash
# round 2, GPU call A: new backward kernel variants - correctness, accuracy at the corners, A/B timing against the generic variant
mkdir -p gpurun_out/r2a
cd /root/repo
python -m pytest tests/test_gpu_sosfilt.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2a/pytest_sos.log
python scripts/eq_accuracy.py > gpurun_out/r2a/eq_accuracy.log 2>&1
for v in "" "DASP_DESIGNED=1" "DASP_DESIGNED=1 DASP_NOGX=1" "DASP_NOGC=1"; do
  echo "== $v" >> gpurun_out/r2a/sosbench.log
  env DASP_PEQ=1 $v ./tools/sosbench 256 2 131072 400 >> gpurun_out/r2a/sosbench.log 2>&1
  env DASP_PEQ=1 $v ./tools/sosbench 256 2 131072 400 >> gpurun_out/r2a/sosbench.log 2>&1
done
python bench.py --no-secondary --no-cpu-baseline --steps 300 > gpurun_out/r2a/bench_eager.json 2> gpurun_out/r2a/bench_eager.err
tail -3 gpurun_out/r2a/pytest_sos.log; cat gpurun_out/r2a/eq_accuracy.log; cat gpurun_out/r2a/sosbench.log; cat gpurun_out/r2a/bench_eager.json
