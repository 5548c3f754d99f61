#!/bin/bash
# Developer helper: variant of libdasp_hip.so with extra -D flags on EVERY source (flags that live in common.hpp).
# usage: scripts/build_variant_all.sh <name> [-DFLAG=VALUE ...] -> tools/<name>/libdasp_hip.so
set -e
name=$1; shift
mkdir -p tools/$name
c=dasp_pytorch_amd/csrc
pids=()
for src in $c/*.hip; do
  b=$(basename $src .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-inline-asm -Wno-pass-failed "$@" -c $src -o tools/$name/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libdasp_hip.so -o tools/$name/libdasp_hip.so tools/$name/*.o
echo tools/$name/libdasp_hip.so
