#!/bin/bash
# round 6: upper bounds for the filter-bank options (timing-only variant libraries), same box, interleaved
out=gpurun_out/r06/fb_bounds.log; mkdir -p gpurun_out/r06; : > $out
for i in 1 2 3; do
  for v in in-tree fb_bound1 fb_bound2; do
    if [ $v = in-tree ]; then lib=""; else lib=$GRAFT_REPO_ROOT/tools/$v/libdasp_hip.so; fi
    echo "== $v" >> $out
    DASP_HIP_LIB=$lib DASP_RV_NOISE=generated python scripts/reverb_time.py 128 2 262144 2>/dev/null | tail -1 | cut -c1-200 >> $out
    DASP_HIP_LIB=$lib DASP_RV_NOISE=generated python scripts/reverb_time.py 16 1 131072 2>/dev/null | tail -1 | cut -c1-200 >> $out
  done
done
cat $out
