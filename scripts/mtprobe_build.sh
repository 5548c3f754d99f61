#!/bin/bash
# usage: scripts/mtprobe_build.sh NAME "-DFLAG=.. ..."   -> tools/mtprobe/NAME.so (the library with csrc/mtrand.hip compiled with the flags)
set -e
cd "$(dirname "$0")/.."
C=dasp_pytorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-pass-failed -Wno-inline-asm $2 -c $C/mtrand.hip -o tools/mtprobe/$1.o
objs=$(ls $C/*.o | grep -v mtrand.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mtprobe/$1.so $objs tools/mtprobe/$1.o
rm tools/mtprobe/$1.o
