#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_d
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "collect or facade or board_eval or render or canaries or c1_thousand" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python scripts/sweep_sizes.py --sizes 4096,131072,262144,1048576 --modes full,traj,mask,trajmask --plies 320 > $O/sweep.jsonl
cat $O/sweep.jsonl
python scripts/bench_facade.py > $O/facade.txt 2>&1; tail -2 $O/facade.txt
python -c "
import cProfile, pstats, sys, io
sys.argv=['bench_facade.py','--games','100']
import runpy
pr=cProfile.Profile(); pr.enable()
runpy.run_path('scripts/bench_facade.py', run_name='__main__')
pr.disable(); s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:4000])
" > $O/facade_profile.txt 2>&1; head -60 $O/facade_profile.txt
scripts/microbench/winner_lanes > $O/winner_lanes.txt; cat $O/winner_lanes.txt
GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/greedy_stamps.py 65536 > $O/greedy_stamps.txt; cat $O/greedy_stamps.txt
