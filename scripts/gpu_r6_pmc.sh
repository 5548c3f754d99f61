#!/bin/bash
# round 6, PMC passes only (a call of their own, so that the counters' profiling state cannot touch the timed runs of gpu_r6_final.sh):
# HBM bytes of the EQ kernels and of the compressor / expander / gain / distortion kernels, hash-tied to the sources that were measured
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
bash scripts/hbm_traffic.sh $out > $out/hbm_traffic.log 2>&1; tail -c 300 $out/hbm_traffic.log; echo
timeout 900 bash scripts/ops_traffic.sh $out/hbm_traffic_ops.json 2>&1 | tail -2
rm -rf gpurun_out/pmc_r2
