#!/bin/bash
# A/B builds: scripts/build_variant.sh <name> [extra hipcc flags] -> ab/libpco_gfx_<name>.so (use with PCO_GFX_LIB=...)
name=$1; shift
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-result "$@" pcodec_amd/csrc/pco_gfx.hip -o ab/libpco_gfx_$name.so
