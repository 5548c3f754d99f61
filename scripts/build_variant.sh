#!/bin/bash
# Developer helper: a variant build of libdasp_hip.so for A/B runs. usage: scripts/build_variant.sh <name> [-DFLAG=VALUE ...]
# -> tools/<name>/libdasp_hip.so; run with LD_LIBRARY_PATH=tools/<name> ./tools/sosbench ... or DASP_HIP_LIB=tools/<name>/libdasp_hip.so python ...
set -e
name=$1; shift
mkdir -p tools/$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-inline-asm -Wno-pass-failed -shared "$@" dasp_pytorch_amd/csrc/*.hip -o tools/$name/libdasp_hip.so
echo tools/$name/libdasp_hip.so
