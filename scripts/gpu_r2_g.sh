#!/bin/bash
# round 2, GPU call G: the profile set for profiles/r02 - bench line, rocprof kernel stats of the same command, HBM traffic counters, smoke
mkdir -p gpurun_out/r02
cd /root/repo
export TMPDIR=/tmp
bash scripts/hbm_traffic.sh gpurun_out/r02 > gpurun_out/r02/hbm_traffic.log 2>&1
mkdir -p profiles/r02 && cp gpurun_out/r02/hbm_traffic.json profiles/r02/hbm_traffic.json
python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof -o p -- python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r02/bench_under_rocprof.json 2> gpurun_out/r02/prof.err
cp $(find gpurun_out/r02/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r02/bench_kernel_stats.csv
rm -rf gpurun_out/r02/prof
python bench.py --launch both --no-secondary --no-cpu-baseline > gpurun_out/r02/bench_both.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02/smoke.log 2>&1; tail -3 gpurun_out/r02/smoke.log
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'roofline', 'roofline_fwd', 'small_kernels_ms', 'traffic_file')})
print(d['cpu_baseline'])
for k, v in d['secondary'].items():
    print(k, v['shape'], 'wall', v['ms_fwd_bwd'], 'gpu', v.get('gpu_ms_fwd_bwd'))
print(json.load(open('gpurun_out/r02/bench_both.json'))['launch_ms_per_step'])
PY
head -6 gpurun_out/r02/bench_kernel_stats.csv | cut -c1-200
