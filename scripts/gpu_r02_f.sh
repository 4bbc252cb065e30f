#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_f
rm -rf $O && mkdir -p $O
for m in traj trajmask; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES -d $O/${m}_sq1 -o p -- python3 scripts/run_eager.py $m 1048576 6 > $O/${m}_sq1.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/${m}_sq2 -o p -- python3 scripts/run_eager.py $m 1048576 6 > $O/${m}_sq2.log 2>&1
  python scripts/rocpd_summary.py counters k_collect $O/${m}_sq1/p_results.db $O/${m}_sq2/p_results.db > $O/${m}_counters.csv
  cat $O/${m}_counters.csv
done
rocprofv3 --kernel-trace -d $O/trace -o p -- python3 scripts/run_eager.py traj 1048576 6 > $O/trace.log 2>&1
python scripts/rocpd_summary.py bygrid k_collect $O/trace/p_results.db 1
