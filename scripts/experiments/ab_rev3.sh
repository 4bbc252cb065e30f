#!/bin/bash
# gbl_greedy / gbl_collect_policy between libraries built beforehand (scripts/experiments/build_rev.sh / build_variant.sh: no git on the GPU box)
L=${LIBS:-"build/lib_r3.so build/lib_head.so build/lib_cur.so"}
for n in ${SIZES:-4096 65536 1048576}; do python scripts/ab_greedy.py $n $L 2>&1 | grep -v amdgpu.ids | cut -c1-100; done
for n in ${PSIZES:-65536}; do python scripts/ab_policy_collect.py $n 16 $L 2>&1 | grep -v amdgpu.ids | cut -c1-100; done
