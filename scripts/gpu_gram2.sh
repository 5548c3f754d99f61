#!/bin/bash
# Gram-matrix backward: timing of variant builds with tools/sosbench (bwd + finalize at the north-star shape), phase stamps of one wave
out=gpurun_out/gram2; mkdir -p $out; log=$out/variants.log; : > $log
run() { echo "== $1" >> $log; shift; env "$@" DASP_PEQ=1 DASP_DESIGNED=1 ./tools/sosbench 256 2 131072 300 2>&1 | grep -v "^check\|^prep phases\|^  phi\|^trace\|^section\|^bwd tile" >> $log; }
run "old kernel" DASP_BWD_GRAM=0
run "gram (in-tree)" DASP_BWD_GRAM=1
for v in "$@"; do run "gram $v" LD_LIBRARY_PATH=tools/gram_$v DASP_TRACE=1; done
run "gram (in-tree), no gx" DASP_BWD_GRAM=1 DASP_NOGX=1
run "gram (in-tree)" DASP_BWD_GRAM=1
cat $log
