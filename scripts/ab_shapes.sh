#!/bin/bash
# Sweep of the greedy workgroup shapes by batch size, gbl_greedy and gbl_collect_policy (the tables above greedy_shape() /
# policy_shape() in gobblet_hip.hip).  Build one library per shape first (shape code = what GBL_FORCE_GREEDY_SHAPE takes:
# 14 <1,4>, 18 <1,8>, 26 <1,16>, 28 <2,8>, 48 <4,8>, 56 <4,16>), then run this on the GPU box:
#   for s in 14 18 26 28 48 56; do scripts/build_variant.sh s$s -DGBL_FORCE_GREEDY_SHAPE=$s; done
#   gpurun -- 'bash scripts/ab_shapes.sh'
L="build/lib_s14.so build/lib_s18.so build/lib_s26.so build/lib_s28.so build/lib_s48.so build/lib_s56.so"
for n in 4096 16384 32768 49152 65536 98304 131072 196608 262144 1048576; do
  timeout -k 10 120 python scripts/ab_greedy.py $n $L >> gpurun_out/r3_shapes_greedy.txt 2>&1 || { tail -5 gpurun_out/r3_shapes_greedy.txt; exit 1; }
done
for n in 4096 16384 32768 65536 131072 262144; do
  timeout -k 10 160 python scripts/ab_policy_collect.py $n 16 $L >> gpurun_out/r3_shapes_policy.txt 2>&1 || { tail -5 gpurun_out/r3_shapes_policy.txt; exit 1; }
done
grep -v amdgpu.ids gpurun_out/r3_shapes_greedy.txt | cut -c1-62
grep -v amdgpu.ids gpurun_out/r3_shapes_policy.txt | cut -c1-72
