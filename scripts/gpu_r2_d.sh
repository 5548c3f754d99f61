#!/bin/bash
# round 2, GPU call D: segmented compressor tests, small-batch timings, HBM traffic counters for HEAD
mkdir -p gpurun_out/r2d
cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_gpu_dynamics.py tests/test_gpu_sosfilt.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r2d/pytest.log; tail -5 gpurun_out/r2d/pytest.log
bash scripts/hbm_traffic.sh gpurun_out/r2d 2>&1 | tail -2
cp gpurun_out/r2d/hbm_traffic.json profiles/r02/hbm_traffic.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 100 > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; tail -2 gpurun_out/r2d/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r2d/bench.json'))
print(d['ms_per_step'], d['roofline']['traffic'], d['traffic_file'])
for k, v in d['secondary'].items():
    print(k, v['shape'], 'wall', v['ms_fwd_bwd'], 'gpu', v.get('gpu_ms_fwd_bwd'))
PY
