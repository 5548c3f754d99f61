#!/bin/bash
# A/B of variant library builds on one box: scripts/gpu_ab.sh <out> <variant dir or "head"> ...   (interleaved, 3 rounds)
out=$1; shift
mkdir -p gpurun_out/$out
cd /root/repo
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = head ]; then lp=dasp_pytorch_amd/csrc; else lp=tools/$v; fi
    echo -n "$v: " >> gpurun_out/$out/ab.log
    LD_LIBRARY_PATH=$lp DASP_PEQ=1 DASP_DESIGNED=1 ./tools/sosbench 256 2 131072 400 2>&1 | grep shape | sed 's/shape (256,2,131072) S=6: //' >> gpurun_out/$out/ab.log
  done
done
cat gpurun_out/$out/ab.log
