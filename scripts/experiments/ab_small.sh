#!/bin/bash
# k_collect_small against the kernels it replaces on small batches, in one process per size (scripts/ab_inproc.py):
#   T plies per launch with every ply materialised (FULL, MASK_ONLY) and the one-ply pipeline (gbl_rollout, plies = 1).
set -e -o pipefail
O=${1:-gpurun_out/ab_small}
mkdir -p $O
scripts/build_variant.sh small0 -DGBL_FORCE_COLLECT_SMALL=0 > /dev/null
scripts/build_variant.sh small1 -DGBL_FORCE_COLLECT_SMALL=4 > /dev/null
for n in ${SIZES:-1024 4096 8192 16384 32768}; do
  python scripts/ab_inproc.py $n 32 all build/lib_small0.so build/lib_small1.so
  python scripts/ab_inproc.py $n 32 mask build/lib_small0.so build/lib_small1.so
  python scripts/ab_inproc.py $n 1 ply build/lib_small0.so build/lib_small1.so
done 2>&1 | grep -v "^$" | tee $O/ab_small.txt
