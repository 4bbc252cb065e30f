#!/bin/bash
# Round-2 opening measurement (one GPU call): instruction issue costs + kernel-only durations per batch size
# of the round-1 kernels.   gpurun -- 'scripts/gpu_baseline_r02.sh'
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_base
rm -rf $O && mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
scripts/microbench/valu_rates > $O/valu_rates.txt
cat $O/valu_rates.txt
python scripts/sweep_sizes.py --sizes 4096,65536,131072,262144,524288,1048576 > $O/sweep.jsonl
cat $O/sweep.jsonl
rocprofv3 --kernel-trace -d $O/trace -o p -- python3 scripts/sweep_sizes.py --sizes 4096,65536,131072,262144,524288,1048576 --plies 100 --reps 2 > $O/trace.log 2>&1
python scripts/rocpd_summary.py bygrid k_rollout $O/trace/p_results.db 64 > $O/bygrid.csv
cat $O/bygrid.csv
