#!/bin/bash
# round 4, A/B 3: forward cascade on the matrix cores - everywhere (in-tree), one-workgroup-per-row launches only (mo1), nowhere (mo0)
out=gpurun_out/r4_ab3.log; : > $out
for rep in 1 2; do
  for v in in-tree mo1 mo0; do
    if [ $v = in-tree ]; then unset DASP_HIP_LIB; else export DASP_HIP_LIB=$PWD/tools/$v/libdasp_hip.so; fi
    python scripts/small_batch_graph.py $v 2>/dev/null >> $out
  done
done
unset DASP_HIP_LIB
python scripts/chain_fwd_ab.py final >> $out 2>/dev/null
