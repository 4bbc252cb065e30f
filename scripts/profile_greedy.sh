#!/bin/bash
# Config 5 (greedy) measurements and rocprofv3 passes -> gpurun_out/c5/ ; then
#   python scripts/profile_collect.py r01 greedy
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/c5
rm -rf $O && mkdir -p $O
python scripts/bench_greedy.py > $O/d2.json 2>/dev/null
python scripts/bench_greedy.py --depth 1 > $O/d1.json 2>/dev/null
python scripts/bench_greedy.py --boards 1048576 --iters 10 > $O/d2_1m.json 2>/dev/null
python scripts/bench_greedy_policy.py 65536 2 > $O/policy.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -o p -- python3 scripts/bench_greedy.py > $O/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $O/pmc1 -o p -- python3 scripts/bench_greedy.py > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc2 -o p -- python3 scripts/bench_greedy.py > $O/pmc2.log 2>&1
cat $O/d2.json
