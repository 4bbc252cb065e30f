#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_i
rm -rf $O && mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-40s value %.3e  ms/step %.5f  frac %.3f  launch_us %.1f" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["mean_launch_us"]))
PY
}
python scripts/sweep_sizes.py --sizes 1048576 --modes traj,full --plies 640 --reps 3 --traj 32 > $O/sweep.jsonl; cut -c1-140 $O/sweep.jsonl
for k in 640 1000 20; do
  python bench.py --steps $k --no-configs --no-cpu-baseline > $O/bench_k$k.json; show $O/bench_k$k.json
done
python bench.py --steps 640 --graph 0 --no-configs --no-cpu-baseline > $O/bench_k640_eager.json; show $O/bench_k640_eager.json
python bench.py --steps 640 --traj 64 --no-configs --no-cpu-baseline > $O/bench_k640_T64.json; show $O/bench_k640_T64.json
python bench.py --steps 640 --traj 8 --no-configs --no-cpu-baseline > $O/bench_k640_T8.json; show $O/bench_k640_T8.json
python scripts/sweep_sizes.py --sizes 1048576 --modes traj,full --plies 640 --reps 3 --traj 32 >> $O/sweep.jsonl; tail -2 $O/sweep.jsonl | cut -c1-140
