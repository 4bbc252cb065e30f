#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_b
rm -rf $O && mkdir -p $O
scripts/microbench/valu_rates > $O/valu_rates.txt
cat $O/valu_rates.txt
for n in 4096 131072 262144 1048576; do
  GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/phase_stamps.py $n > $O/stamps_$n.txt
  cat $O/stamps_$n.txt
done
