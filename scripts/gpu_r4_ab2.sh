#!/bin/bash
# round 4, A/B 2: packed (v_pk_*) against scalar complex arithmetic in fft_lds.hpp - reverb and STFT-loss kernels, same box, interleaved
out=gpurun_out/r4_ab2.log; : > $out
for rep in 1 2 3; do
  for v in in-tree scalarfft; do
    if [ $v = in-tree ]; then unset DASP_HIP_LIB; else export DASP_HIP_LIB=$PWD/tools/$v/libdasp_hip.so; fi
    python scripts/small_batch_graph.py $v 2>/dev/null >> $out
    python scripts/loss_time.py 2>/dev/null | sed "s/^/$v /" >> $out
  done
done
