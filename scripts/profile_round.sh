#!/bin/bash
# Everything the numbers in README.md / DESIGN.md / profiles/ come from, in one GPU call:
#   gpurun -- 'scripts/profile_round.sh'      then      python scripts/profile_collect.py r01
# Writes bench lines and rocprofv3 databases under gpurun_out/final/.
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python bench.py > $O/bench_default.json
python bench.py --no-obs --no-cpu-baseline > $O/bench_maskonly.json
python bench.py --mode step --no-cpu-baseline > $O/bench_stepmode.json
python bench.py --boards 4194304 --steps 300 --no-cpu-baseline > $O/bench_4194304_boards.json
python bench.py --boards 2097152 --steps 500 --no-cpu-baseline > $O/bench_2097152_boards.json
python bench.py --boards 262144 --no-cpu-baseline > $O/bench_c3_262144_boards.json
python bench.py --boards 4096 --no-cpu-baseline > $O/bench_c2_4096_boards.json
python scripts/bench_playouts.py > $O/playouts.json 2> /dev/null
echo "bench lines done"
rocprofv3 --kernel-trace --stats -d $O/fused_stats -o p -- python3 bench.py --steps 300 --no-cpu-baseline > $O/fused_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/step_stats -o p -- python3 bench.py --mode step --steps 300 --no-cpu-baseline > $O/step_stats.log 2>&1
echo "kernel traces done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/fused_pmc_$c -o p -- python3 bench.py --steps 20 --warmup 8 --graph 0 --no-cpu-baseline > $O/fused_pmc_$c.log 2>&1
  rocprofv3 --pmc $c -d $O/step_pmc_$c -o p -- python3 bench.py --mode step --steps 20 --warmup 8 --graph 0 --no-cpu-baseline > $O/step_pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES -d $O/fused_sq1 -o p -- python3 bench.py --steps 20 --warmup 8 --graph 0 --no-cpu-baseline > $O/fused_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/fused_sq2 -o p -- python3 bench.py --steps 20 --warmup 8 --graph 0 --no-cpu-baseline > $O/fused_sq2.log 2>&1
echo "counters done"
ls $O
