#!/bin/bash
# Developer helper: variant of libdasp_hip.so that differs from the in-tree build only in ONE source's -D flags (the other objects are
# reused). usage: scripts/build_variant_one.sh <name> <source.hip> [-DFLAG=VALUE ...] -> tools/<name>/libdasp_hip.so
set -e
name=$1; src=$2; shift; shift
mkdir -p tools/$name
c=dasp_pytorch_amd/csrc
b=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-inline-asm -Wno-pass-failed "$@" -c $c/$b.hip -o tools/$name/$b.o
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libdasp_hip.so -o tools/$name/libdasp_hip.so tools/$name/$b.o $(ls $c/*.o | grep -v "/$b.o")
echo tools/$name/libdasp_hip.so
