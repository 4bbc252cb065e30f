#!/bin/bash
# Why does gbl_collect run at 0.81 of the HBM peak at 2^22 boards when 2^20 boards reach 0.88 and the one-ply kernel 0.887 at 2^22?
# Counters, not sweeps: separate rocprofv3 --pmc passes -- the L2's write-request interface to the fabric
# (requests, stalls, credit stalls, requests in flight), its per-channel spread, and the address-translation path (UTCL1 hits /
# misses / stalls, UTCL2 busy) -- for k_collect (T = 8) at 2^20 and 2^22 boards and k_rollout (one ply) at 2^22.
#   gpurun -- 'scripts/experiments/pmc_large.sh [outdir]'    then the summaries are under outdir/*.counters.csv
set -e -o pipefail
export TMPDIR=/tmp
O=${1:-gpurun_out/pmc_large}
rm -rf $O && mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1   # (no compiler may run under the profiler's preload)
P1="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
P2="TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
P3="TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL"
P4="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
P5="TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_PENDING_STALL_CYCLES_sum"
P6="GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
P7="TCC_CYCLE_sum TCC_IB_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_WRITE_sum"
while read name mode boards launches T; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7"; do
    i=$((i+1))
    rocprofv3 --pmc $P -d $O/${name}_p$i -o p -- python3 scripts/run_eager.py $mode $boards $launches $T > $O/${name}_p$i.log 2>&1 || echo "pass $i of $name failed" >> $O/failed.txt
  done
  echo "$name done"
done <<EOT
collect_1048576_T8 traj 1048576 4 8
collect_4194304_T8 traj 4194304 3 8
collect_4194304_T2 traj 4194304 6 2
rollout_4194304 full 4194304 6 1
EOT
python scripts/experiments/pmc_large_reduce.py $O
