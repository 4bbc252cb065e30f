#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_j
rm -rf $O && mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-34s value %.3e  ms/step %.5f  frac %.3f  launch_us %.1f  T %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["mean_launch_us"], d["config"].get("plies_per_launch")))
for k, v in d.get("configs", {}).items():
    print("    %-30s %.3e %s  us/step %.3f  frac %s" % (k, v["value"], v["unit"], v["us_per_step"], v["roofline"]["frac"]))
PY
}
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json; show $O/bench_driver.json
python bench.py --steps 20 --warmup 5 --traj 4 --no-configs --no-cpu-baseline > $O/bench_k20_T4.json; show $O/bench_k20_T4.json
python bench.py --steps 20 --warmup 5 --traj 10 --no-configs --no-cpu-baseline > $O/bench_k20_T10.json; show $O/bench_k20_T10.json
python bench.py --steps 20 --warmup 5 --mode fused --no-configs --no-cpu-baseline > $O/bench_k20_fused.json; show $O/bench_k20_fused.json
python bench.py --no-configs --no-cpu-baseline > $O/bench_k1000.json; show $O/bench_k1000.json
python bench.py --boards 131072 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_131072_k20.json; show $O/bench_131072_k20.json
python bench.py --boards 131072 --no-configs --no-cpu-baseline > $O/bench_131072.json; show $O/bench_131072.json
