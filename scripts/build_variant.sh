#!/bin/bash
# Build an EXPERIMENT library of the same ABI for A/B runs (-DGBL_AB_BUILD: the knobs of gobblet-rl_amd/csrc/gobblet_ab.h; the product build has none). The scripts under scripts/ load it when GOBBLET_HIP_LIB=build/lib_NAME.so is set:
#   scripts/build_variant.sh NAME [-DGBL_... ...]
set -e
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mcode-object-version=5 \
  -mllvm -amdgpu-kernarg-preload-count=16 -DGBL_AB_BUILD "$@" -o build/lib_$name.so gobblet-rl_amd/csrc/gobblet_hip.hip 2>&1 | grep -v "argument unused" || true
ls -la build/lib_$name.so
