#!/bin/bash
# same-box A/B of the headline kernels: the round-5 library (tools/r05: built from commit db1df0e) against the in-tree library,
# tools/sosbench at the north-star shape, interleaved. usage (GPU box): bash scripts/ab_r05_r06_sosbench.sh
out=gpurun_out/r06/ab_r05_r06_sosbench.log; mkdir -p gpurun_out/r06; : > $out
for i in 1 2 3 4; do
  echo "== r05 (run $i)" >> $out; DASP_PEQ=1 DASP_DESIGNED=1 ./tools/r05/sosbench 256 2 131072 300 2>&1 | head -1 >> $out
  echo "== r06 (run $i)" >> $out; DASP_PEQ=1 ./tools/sosbench 256 2 131072 300 2>&1 | head -1 >> $out
done
cat $out
