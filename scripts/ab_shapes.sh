L="build/lib_s14.so build/lib_s18.so build/lib_s26.so build/lib_s28.so build/lib_s48.so build/lib_s56.so"
for n in 4096 16384 32768 49152 65536 98304 131072 196608 262144 1048576; do
  timeout -k 10 120 python scripts/ab_greedy.py $n $L >> gpurun_out/r3_shapes_greedy.txt 2>&1 || { tail -5 gpurun_out/r3_shapes_greedy.txt; exit 1; }
done
for n in 4096 16384 32768 65536 131072 262144; do
  timeout -k 10 160 python scripts/ab_policy_collect.py $n 16 $L >> gpurun_out/r3_shapes_policy.txt 2>&1 || { tail -5 gpurun_out/r3_shapes_policy.txt; exit 1; }
done
grep -v amdgpu.ids gpurun_out/r3_shapes_greedy.txt | cut -c1-62
grep -v amdgpu.ids gpurun_out/r3_shapes_policy.txt | cut -c1-72
