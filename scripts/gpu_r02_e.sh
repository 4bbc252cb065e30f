#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_e
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "greedy or canaries or abi" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python scripts/bench_greedy.py > $O/greedy_65536.json; cat $O/greedy_65536.json
python scripts/bench_greedy.py --boards 1048576 > $O/greedy_1m.json; cat $O/greedy_1m.json
python scripts/bench_greedy_policy.py > $O/greedy_policy.json 2>&1; tail -2 $O/greedy_policy.json
GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/greedy_stamps.py 65536 > $O/greedy_stamps.txt; cat $O/greedy_stamps.txt
