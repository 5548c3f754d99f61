#!/bin/bash
# round 6: f2-with-gradients A/B (fused EQ -> compressor forward that saves for backward), PMC traffic of the EQ kernels on this round's
# sources, the reference timed on the FULL (256,2,131072) workload
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
for shape in "256 2 131072" "128 2 131072" "256 1 131072"; do timeout 300 python scripts/chain_fwd_saving_ab.py $shape 2>&1 | tail -1 | tee -a $out/chain_fwd_saving_ab.log; done
bash scripts/hbm_traffic.sh $out > $out/hbm_traffic.log 2>&1; tail -c 600 $out/hbm_traffic.log
timeout 1500 python bench.py --steps 20 --warmup 5 --no-secondary --cpu-baseline-full > $out/bench_cpu_full.json 2> $out/bench_cpu_full.err; python -c "
import json; d=json.loads(open('$out/bench_cpu_full.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['cpu_baseline'])"
