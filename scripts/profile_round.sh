#!/bin/bash
# Everything the numbers in README.md / DESIGN.md / profiles/ come from, in one GPU call:
#   gpurun -- 'scripts/profile_round.sh'      then      python scripts/profile_collect.py r04
# Writes bench lines and rocprofv3 databases under gpurun_out/final/.
#   (two calls when one does not fit gpurun's time limit:  scripts/profile_round.sh a   then   scripts/profile_round.sh b;
#    after a kernel change a third, scripts/profile_round.sh c, re-takes the bench lines against the fresh counters)
set -e -o pipefail
export TMPDIR=/tmp
STAGE=${1:-all}
O=gpurun_out/final
[ "$STAGE" = b ] || [ "$STAGE" = c ] || rm -rf $O
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build_$STAGE.log 2>&1   # (no compiler may run under the profiler's preload)
for m in valu_rates valu_mix icache_cold winner_lanes write_classes wave_placement flag_sync dpp_scan; do   # the microbenchmarks this script runs
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/$m scripts/microbench/$m.hip >> $O/build_$STAGE.log 2>&1   # always rebuilt: a stale binary must never publish numbers
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gobblet-rl_amd/csrc -o scripts/microbench/reply_rate scripts/microbench/reply_rate.hip >> $O/build_$STAGE.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I gobblet-rl_amd/csrc -o scripts/microbench/quad_split scripts/microbench/quad_split.hip >> $O/build_$STAGE.log 2>&1
[ "$STAGE" = b ] || scripts/build_variant.sh stamps -DGBL_STAMPS -DGBL_AB_COLLECT_CFG >> $O/build_$STAGE.log 2>&1   # (diagnostic build for the phase stamps)
bench_lines() {
# stdout of bench.py is the compact contract line; the full record (sub-records, per-rank lists) goes to --configs-out
python bench.py --configs-out $O/bench_default_full.json > $O/bench_default.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --configs-out $O/bench_driver_cmd_full.json > $O/bench_driver_cmd.json   # the driver's exact command
python bench.py --mode fused --no-configs --no-cpu-baseline --configs-out $O/bench_single_ply_full.json > $O/bench_single_ply.json
python bench.py --mode step --no-configs --no-cpu-baseline --configs-out $O/bench_stepmode_full.json > $O/bench_stepmode.json
python bench.py --mode step2 --no-configs --no-cpu-baseline --configs-out $O/bench_step_two_launch_full.json > $O/bench_step_two_launch.json
python bench.py --no-obs --no-configs --no-cpu-baseline --configs-out $O/bench_maskonly_full.json > $O/bench_maskonly.json
python bench.py --boards 131072 --no-configs --no-cpu-baseline --configs-out $O/bench_c4_shard_131072_full.json > $O/bench_c4_shard_131072.json
}
if [ "$STAGE" = c ]; then
# the bench lines once more, AFTER scripts/profile_collect.py has written profiles/pmc_traffic.json for these kernel sources
# (a line taken before stage b's counters exist says "traffic": null with the reason); then profile_collect.py again
bench_lines
ls $O/bench_*.json
exit 0
fi
if [ "$STAGE" != b ]; then
# ---- bench lines --------------------------------------------------------------------------------------------
bench_lines
python scripts/bench_greedy.py > $O/greedy_65536.json
python scripts/bench_greedy.py --boards 1048576 > $O/greedy_1048576.json
python scripts/bench_greedy_policy.py > $O/greedy_policy.json 2> /dev/null
python scripts/bench_playouts.py > $O/playouts.json 2> /dev/null
python scripts/bench_facade.py > $O/facade.txt
scripts/microbench/valu_rates > $O/valu_rates.txt
scripts/microbench/valu_mix > $O/valu_mix.txt
scripts/microbench/icache_cold > $O/icache_cold.txt
scripts/microbench/reply_rate > $O/reply_rate.txt
GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/greedy_wave_stamps.py 65536 2> /dev/null > $O/greedy_wave_stamps.txt
GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/greedy_wave_stamps.py 1048576 2> /dev/null >> $O/greedy_wave_stamps.txt
GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/policy_wave_stamps.py 2> /dev/null > $O/policy_wave_stamps.txt || true
for c in 220 120; do AB_LIB=build/lib_stamps.so python scripts/microbench/role_phase_stamps.py 4096 $c; done 2> /dev/null > $O/role_phase_stamps.txt || true
scripts/microbench/winner_lanes > $O/winner_lanes.txt
scripts/microbench/quad_split > $O/quad_split.txt
timeout -k 5 60 scripts/microbench/flag_sync > $O/flag_sync.txt
scripts/microbench/dpp_scan >> $O/flag_sync.txt
scripts/microbench/wave_placement 1024 28672 > $O/wave_placement.txt
if [ -z "$SHORT" ]; then
# ---- placement of the trajectory arrays -------------------------------------------------------
scripts/microbench/write_classes 160 > $O/write_classes.txt
scripts/microbench/write_classes 160 131072 32 | grep -v "^  policy" >> $O/write_classes.txt
python scripts/placement_probe_check.py 224 > $O/placement_probe_check.txt 2> /dev/null
python scripts/placement_probe_check.py 224 131072 32 >> $O/placement_probe_check.txt 2> /dev/null
for i in 1 2 3 4 5; do   # fresh processes: placed by the probe / as the allocator hands the arrays out
  for pl in auto any; do
    python bench.py --no-configs --no-cpu-baseline --placement $pl --configs-out $O/placement_run.json > $O/placement_run.line 2> /dev/null
    python -c "
import json
d = json.loads(open('$O/placement_run.line').read().strip().splitlines()[-1])
f = json.load(open('$O/placement_run.json'))
print('run $i placement $pl: %.3e env-steps/s, %.2f us per ply, roofline.frac %.3f, %s' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['frac'], json.dumps(f['detail']['trajectory_placement_per_rank'][0])))" >> $O/placement_ab.txt
  done
done
# the same with 200 GiB of the device already taken by the process (a trainer's model and replay buffer): the search is capped
# by a quarter of what is free
for i in 1 2 3; do
  python scripts/placement_full_device.py 200 >> $O/placement_ab.txt 2> /dev/null
done
fi   # SHORT
python scripts/soak_parity.py ${SOAK_SECONDS:-150} 3 > $O/soak_parity.txt 2>&1
echo "bench lines done"
# ---- kernel traces (durations) ------------------------------------------------------------------------------
rocprofv3 --kernel-trace --stats -d $O/collect_stats -o p -- python3 bench.py --steps 320 --no-configs --no-cpu-baseline > $O/collect_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/single_stats -o p -- python3 bench.py --mode fused --steps 300 --no-configs --no-cpu-baseline > $O/single_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/step_stats -o p -- python3 bench.py --mode step --steps 300 --no-configs --no-cpu-baseline > $O/step_stats.log 2>&1
rocprofv3 --kernel-trace -d $O/sweep_trace -o p -- python3 scripts/sweep_sizes.py --sizes 4096,16384,32768,65536,131072,262144,1048576 --modes full,mask,traj,trajmask --plies 128 --reps 2 > $O/sweep_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/greedy_stats -o p -- python3 scripts/run_eager.py greedy 65536 20 > $O/greedy_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/policy_stats -o p -- python3 scripts/run_eager.py policy 65536 8 16 > $O/policy_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/driver_stats -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/driver_stats.log 2>&1
echo "kernel traces done"
fi
if [ "$STAGE" != a ]; then
# ---- HBM traffic (separate --pmc passes; eager launches so that counters are attributed per dispatch) ----------
# one line per bench.py record: "run-name mode boards launches plies-per-launch" (profile_collect.py maps them to the keys of
# profiles/pmc_traffic.json: the headline at 8 plies per launch and at the driver's single 20-ply launch, the C2 / C3 / C4-shard /
# 2^22 / MASK_ONLY records, the one-ply kernels at every size bench.py reports, k_step)
cat > $O/pmc_runs.txt <<EOT
collect_T8 traj 1048576 6 8
collect_T20 traj 1048576 4 20
collect_4096_T1024 traj 4096 3 1024
collect_16384_T512 traj 16384 3 512
collect_32768_T256 traj 32768 4 256
collect_65536_T256 traj 65536 4 256
collect_4194304_T4 traj 4194304 4 4
collect_131072_T128 traj 131072 4 128
collect_131072_T20 traj 131072 6 20
collect_262144_T64 traj 262144 4 64
collect_262144_T20 traj 262144 6 20
collect_524288_T32 traj 524288 4 32
collect_524288_T20 traj 524288 4 20
collect_4194304_T8 traj 4194304 4 8
collect_noobs_T16 trajmask 1048576 4 16
fused_1048576 full 1048576 20 1
fused_262144 full 262144 20 1
fused_131072 full 131072 20 1
fused_4096 full 4096 20 1
fused_4194304 full 4194304 10 1
fused_noobs_1048576 mask 1048576 20 1
step_1048576 step 1048576 12 1
step_131072 step 131072 20 1
step2_1048576 step2 1048576 12 1
EOT
while read name mode boards launches T; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_${name}_$c -o p -- python3 scripts/run_eager.py $mode $boards $launches $T > $O/pmc_${name}_$c.log 2>&1
  done
done < $O/pmc_runs.txt
echo "traffic counters done"
# ---- SQ counters ------------------------------------------------------------------------------------------------
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for m in traj trajmask full mask; do
  rocprofv3 --pmc $SQ1 -d $O/${m}_sq1 -o p -- python3 scripts/run_eager.py $m 1048576 6 8 > $O/${m}_sq1.log 2>&1
  rocprofv3 --pmc $SQ2 -d $O/${m}_sq2 -o p -- python3 scripts/run_eager.py $m 1048576 6 8 > $O/${m}_sq2.log 2>&1
done
rocprofv3 --pmc $SQ1 -d $O/greedy_sq1 -o p -- python3 scripts/run_eager.py greedy 65536 20 > $O/greedy_sq1.log 2>&1
rocprofv3 --pmc $SQ2 -d $O/greedy_sq2 -o p -- python3 scripts/run_eager.py greedy 65536 20 > $O/greedy_sq2.log 2>&1
rocprofv3 --pmc $SQ1 -d $O/policy_sq1 -o p -- python3 scripts/run_eager.py policy 65536 6 16 > $O/policy_sq1.log 2>&1
rocprofv3 --pmc $SQ2 -d $O/policy_sq2 -o p -- python3 scripts/run_eager.py policy 65536 6 16 > $O/policy_sq2.log 2>&1
# instruction cache counters of the greedy kernel (~2 000 misses per launch whatever the batch, at no measurable cost)
IC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
rocprofv3 --pmc $IC -d $O/greedy_icache -o p -- python3 scripts/run_eager.py greedy 65536 20 > $O/greedy_icache.log 2>&1
rocprofv3 --pmc $IC -d $O/greedy_icache_1m -o p -- python3 scripts/run_eager.py greedy 1048576 10 > $O/greedy_icache_1m.log 2>&1
echo "counters done"
fi
python scripts/profile_reduce.py   # (databases -> small summaries: gpurun copies at most 64 MiB back)
ls $O
