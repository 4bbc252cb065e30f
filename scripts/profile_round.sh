#!/bin/bash
# Everything the numbers in README.md / DESIGN.md / profiles/ come from, in one GPU call:
#   gpurun -- 'scripts/profile_round.sh'      then      python scripts/profile_collect.py r02
# Writes bench lines and rocprofv3 databases under gpurun_out/final/.
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1   # (no compiler may run under the profiler's preload)
for m in valu_rates winner_lanes write_classes; do   # the microbenchmarks this script runs
  [ scripts/microbench/$m -nt scripts/microbench/$m.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/$m scripts/microbench/$m.hip >> $O/build.log 2>&1
done
# ---- bench lines --------------------------------------------------------------------------------------------
python bench.py > $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_driver_cmd.json
python bench.py --mode fused --no-configs --no-cpu-baseline > $O/bench_single_ply.json
python bench.py --mode step --no-configs --no-cpu-baseline > $O/bench_stepmode.json
python bench.py --no-obs --no-configs --no-cpu-baseline > $O/bench_maskonly.json
python bench.py --boards 131072 --no-configs --no-cpu-baseline > $O/bench_c4_shard_131072.json
python scripts/bench_greedy.py > $O/greedy_65536.json
python scripts/bench_greedy.py --boards 1048576 > $O/greedy_1048576.json
python scripts/bench_greedy_policy.py > $O/greedy_policy.json 2> /dev/null
python scripts/bench_playouts.py > $O/playouts.json 2> /dev/null
python scripts/bench_facade.py > $O/facade.txt
scripts/microbench/valu_rates > $O/valu_rates.txt
scripts/microbench/winner_lanes > $O/winner_lanes.txt
# ---- placement of the trajectory arrays -------------------------------------------------------
scripts/microbench/write_classes 160 > $O/write_classes.txt
scripts/microbench/write_classes 160 131072 32 | grep -v "^  policy" >> $O/write_classes.txt
python scripts/placement_probe_check.py 224 > $O/placement_probe_check.txt 2> /dev/null
python scripts/placement_probe_check.py 224 131072 32 >> $O/placement_probe_check.txt 2> /dev/null
for i in 1 2 3 4 5; do   # fresh processes: placed by the probe / as the allocator hands the arrays out
  for pl in auto any; do
    python bench.py --no-configs --no-cpu-baseline --placement $pl 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i placement $pl: %.3e env-steps/s, %.2f us per ply, roofline.frac %.3f, %s' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['frac'], json.dumps(d['config']['trajectory_placement'])))" >> $O/placement_ab.txt
  done
done
echo "bench lines done"
# ---- kernel traces (durations) ------------------------------------------------------------------------------
rocprofv3 --kernel-trace --stats -d $O/collect_stats -o p -- python3 bench.py --steps 320 --no-configs --no-cpu-baseline > $O/collect_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/single_stats -o p -- python3 bench.py --mode fused --steps 300 --no-configs --no-cpu-baseline > $O/single_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/step_stats -o p -- python3 bench.py --mode step --steps 300 --no-configs --no-cpu-baseline > $O/step_stats.log 2>&1
rocprofv3 --kernel-trace -d $O/sweep_trace -o p -- python3 scripts/sweep_sizes.py --sizes 4096,131072,262144,1048576 --modes full,mask,traj,trajmask --plies 128 --reps 2 > $O/sweep_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/greedy_stats -o p -- python3 scripts/run_eager.py greedy 65536 20 > $O/greedy_stats.log 2>&1
echo "kernel traces done"
# ---- HBM traffic (separate --pmc passes; eager launches so that counters are attributed per dispatch) ----------
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/collect_pmc_$c -o p -- python3 scripts/run_eager.py traj 1048576 6 8 > $O/collect_pmc_$c.log 2>&1
  rocprofv3 --pmc $c -d $O/single_pmc_$c -o p -- python3 scripts/run_eager.py full 1048576 20 > $O/single_pmc_$c.log 2>&1
done
# (the shards of BASELINE C4 at 8 / 4 / 2 GPUs, at the plies per launch bench.py picks for --steps 1000 and for --steps 20)
for cfg in "131072 32" "131072 20" "262144 16" "524288 8"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/shard_$1_T$2_pmc_$c -o p -- python3 scripts/run_eager.py traj $1 6 $2 > $O/shard_$1_T$2_pmc_$c.log 2>&1
  done
done
# ---- SQ counters ------------------------------------------------------------------------------------------------
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for m in traj trajmask full mask; do
  rocprofv3 --pmc $SQ1 -d $O/${m}_sq1 -o p -- python3 scripts/run_eager.py $m 1048576 6 8 > $O/${m}_sq1.log 2>&1
  rocprofv3 --pmc $SQ2 -d $O/${m}_sq2 -o p -- python3 scripts/run_eager.py $m 1048576 6 8 > $O/${m}_sq2.log 2>&1
done
rocprofv3 --pmc $SQ1 -d $O/greedy_sq1 -o p -- python3 scripts/run_eager.py greedy 65536 20 > $O/greedy_sq1.log 2>&1
rocprofv3 --pmc $SQ2 -d $O/greedy_sq2 -o p -- python3 scripts/run_eager.py greedy 65536 20 > $O/greedy_sq2.log 2>&1
echo "counters done"
ls $O
