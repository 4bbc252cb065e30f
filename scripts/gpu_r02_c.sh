#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_c
rm -rf $O && mkdir -p $O
for n in 4096 131072 262144 1048576; do
  GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/phase_stamps.py $n > $O/stamps_$n.txt
  cat $O/stamps_$n.txt
done
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -3 $O/gputests.log
python scripts/bench_facade.py > $O/facade.json 2>&1; tail -2 $O/facade.json
