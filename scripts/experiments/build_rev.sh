#!/bin/bash
# Build the library as of a git revision (kernel sources of that revision, today's flags) for in-process A/B runs:
#   scripts/experiments/build_rev.sh NAME REV [-DGBL_... ...]   ->   build/lib_NAME.so
set -e
name=$1; rev=$2; shift; shift
tmp=$(mktemp -d)
mkdir -p $tmp/gobblet-rl_amd/csrc $tmp/include build
for f in gobblet-rl_amd/csrc/gobblet_hip.hip gobblet-rl_amd/csrc/gobblet_device.h gobblet-rl_amd/csrc/gobblet_diag.h include/gobblet_hip.h; do
  git show $rev:$f > $tmp/$f
done
for f in gobblet-rl_amd/csrc/gobblet_knobs.h gobblet-rl_amd/csrc/gobblet_ab.h; do   # (revisions since round 6 have them)
  git show $rev:$f > $tmp/$f 2> /dev/null || rm -f $tmp/$f
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mcode-object-version=5 \
  -mllvm -amdgpu-kernarg-preload-count=16 -DGBL_AB_BUILD "$@" -o build/lib_$name.so $tmp/gobblet-rl_amd/csrc/gobblet_hip.hip 2>&1 | grep -v "argument unused" || true
rm -rf $tmp
ls -la build/lib_$name.so
