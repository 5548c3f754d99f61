#!/bin/bash
# same-box A/B of variant libraries with tools/sosbench at the north-star shape: usage gpu_ab_lib.sh <variant dir name> ...  (in-tree first and last)
out=gpurun_out/ab_lib.log; : > $out
run() { echo "== $1" >> $out; shift; env "$@" DASP_PEQ=1 DASP_DESIGNED=1 ./tools/sosbench 256 2 131072 300 2>&1 | head -1 >> $out; }
for rep in 1 2; do
  run "in-tree" A=1
  for v in "$@"; do run "$v" LD_LIBRARY_PATH=tools/$v; done
done
cat $out
