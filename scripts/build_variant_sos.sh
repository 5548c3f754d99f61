#!/bin/bash
# Developer helper: variant of libdasp_hip.so that differs from the in-tree build only in csrc/sosfilt.hip's -D flags (the other objects are
# reused: one compile of ~25 s instead of all sources). usage: scripts/build_variant_sos.sh <name> [-DFLAG=VALUE ...] -> tools/<name>/libdasp_hip.so
set -e
name=$1; shift
mkdir -p tools/$name
c=dasp_pytorch_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-inline-asm -Wno-pass-failed "$@" -c $c/sosfilt.hip -o tools/$name/sosfilt.o
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libdasp_hip.so -o tools/$name/libdasp_hip.so tools/$name/sosfilt.o $(ls $c/*.o | grep -v sosfilt.o)
echo tools/$name/libdasp_hip.so
