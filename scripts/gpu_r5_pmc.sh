#!/bin/bash
# round 5, call 1 of 2: the PMC passes only (HBM bytes of the EQ kernels, hash-tied to the sources, and of the reverb kernels) - on a box
# of their own, so that the counters' profiling state cannot touch the timed runs of call 2 (scripts/gpu_r5_final.sh skip-pmc)
out=gpurun_out/r05; mkdir -p $out; export TMPDIR=/tmp
bash scripts/hbm_traffic.sh $out > $out/hbm_traffic.log 2>&1; tail -c 400 $out/hbm_traffic.log
DASP_RV_NOISE=generated bash scripts/reverb_traffic.sh $out/hbm_traffic_secondary.json 2>&1 | tail -2
