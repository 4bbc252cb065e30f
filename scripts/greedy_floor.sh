#!/bin/bash
# What a greedy block cannot go below: gbl_greedy (depth 2) with phases of the decision LEFT OUT (-DGBL_X_GREEDY_SKIP=bits: 1 the
# owners' tail, 2 the chunk phase, 4 the list phase, 8 the B phase, 16 the candidate rows; results are wrong, timing only), in one
# process per size against the product build:   gpurun -- 'scripts/greedy_floor.sh [outdir]'
set -e -o pipefail
O=${1:-gpurun_out/greedy_floor}
mkdir -p $O
scripts/build_variant.sh gfull > /dev/null
for b in 1 2 6 14 16 17 31; do scripts/build_variant.sh gskip$b -DGBL_X_GREEDY_SKIP=$b > /dev/null; done
# round 6: the depth-2 pairs of a block capped (-DGBL_X_GREEDY_PAIR_CAP=n; <4,16>: 256 boards, 1 400 pairs = 22 chunks of 64 over 16 wavefronts):
# what ONE pass of chunks (<= 1 024 pairs = 4 per board) would take if some rule removed the rest FOR FREE -- the upper bound of any such rule
for c in ${CAPS:-1024 768 512}; do scripts/build_variant.sh gcap$c -DGBL_X_GREEDY_PAIR_CAP=$c > /dev/null; done
for n in ${SIZES:-4096 65536 1048576}; do
  AB_ALLOW_DIFF=1 python scripts/ab_greedy.py $n build/lib_gfull.so build/lib_gskip1.so build/lib_gskip2.so build/lib_gskip6.so \
      build/lib_gskip14.so build/lib_gskip16.so build/lib_gskip17.so build/lib_gskip31.so $(for c in ${CAPS:-1024 768 512}; do echo build/lib_gcap$c.so; done)
done 2>&1 | grep -v amdgpu.ids | tee $O/greedy_floor.txt
