#!/bin/bash
# The greedy kernels after a change: parity tests, then launch times at the sizes of the shape table.
set -e -o pipefail
O=gpurun_out/${1:-greedy}
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_scan scripts/microbench/dpp_scan.hip 2> /dev/null && /tmp/dpp_scan
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_policy_collect.py -x -q -m gpu -k "greedy or policy" 2>&1 | tail -4
for n in 4096 16384 32768 65536 131072 262144 1048576; do
  python scripts/bench_greedy.py --boards $n 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gbl_greedy %8d boards: %7.2f us per launch, parity %s' % (d['boards'], d['ms_per_call'] * 1e3, d['parity']))"
done | tee $O/greedy_times.txt
for n in 16384 65536; do python scripts/ab_policy_collect.py $n 16 gobblet-rl_amd/csrc/libgobblet_hip.so 2> /dev/null | tail -2; done | tee $O/policy_times.txt
