#!/bin/bash
# The role kernel's forms (k_collect_small<LA, KO, MERGE>: cfg = 100 LA + 10 KO + MERGE) against k_collect2 / k_collect (cfg 0),
# one process per size (scripts/ab_cfg.py): T = 32 plies per launch, every ply materialised, FULL and MASK_ONLY.
#   gpurun -- 'scripts/experiments/ab_roles.sh [outdir]'      SIZES="..." / CFGS="..." override the batch sizes / the forms
set -e -o pipefail
O=${1:-gpurun_out/ab_roles}
mkdir -p $O
scripts/build_variant.sh ab -DGBL_AB_COLLECT_CFG > /dev/null
for n in ${SIZES:-2048 4096 8192 12288 16384 24576 32768 49152 65536 98304 131072}; do
  python scripts/ab_cfg.py $n 32 all ${CFGS:-0 410 210 110 140 141 120 121 111 220 221}
  python scripts/ab_cfg.py $n 32 mask ${MCFGS:-0 410 210 110 411 211 111}
done 2>&1 | grep -v "amdgpu.ids" | tee $O/ab_roles.txt
