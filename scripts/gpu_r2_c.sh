#!/bin/bash
# round 2, GPU call C: bench (eager + graph), rocprof kernel stats of the same command, HBM traffic counters for HEAD
mkdir -p gpurun_out/r2c profiles/r02
cd /root/repo
export TMPDIR=/tmp
python bench.py > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
cat gpurun_out/r2c/bench.json; tail -3 gpurun_out/r2c/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2c/prof -o p -- python bench.py --no-secondary --no-cpu-baseline --steps 100 > gpurun_out/r2c/bench_prof.json 2> gpurun_out/r2c/prof.err
f=$(find gpurun_out/r2c/prof -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -12 "$f"
cp "$f" gpurun_out/r2c/kernel_stats.csv 2>/dev/null
bash scripts/hbm_traffic.sh gpurun_out/r2c 2>&1 | tail -3
