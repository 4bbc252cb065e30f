#!/bin/bash
# The role kernel's three forms (k_collect_small with 4 / 2 / 1 lanes per board) against k_collect2 / k_collect, in one process
# per size (scripts/ab_inproc.py): T = 32 plies per launch, every ply materialised, FULL and MASK_ONLY.
#   gpurun -- 'scripts/ab_roles.sh [outdir]'      SIZES="..." overrides the batch sizes
set -e -o pipefail
O=${1:-gpurun_out/ab_roles}
mkdir -p $O
scripts/build_variant.sh roles0 -DGBL_FORCE_COLLECT_SMALL=0 > /dev/null
for l in 4 2 1; do scripts/build_variant.sh roles$l -DGBL_FORCE_COLLECT_SMALL=$l > /dev/null; done
for n in ${SIZES:-4096 8192 12288 16384 24576 32768 49152 65536 98304 131072}; do
  for st in all mask; do
    python scripts/ab_inproc.py $n 32 $st build/lib_roles0.so build/lib_roles4.so build/lib_roles2.so build/lib_roles1.so
  done
done 2>&1 | grep -v "^$" | tee $O/ab_roles.txt
