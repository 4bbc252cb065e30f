#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_g
rm -rf $O && mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1 || { tail -20 $O/build_smoke.log; exit 1; }
tail -1 $O/build_smoke.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_g/bench_default.json").read().strip().splitlines()[-1])
print("value %.3e  ms/step %.4f  frac %.3f  kernel %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"]))
for k, v in d.get("configs", {}).items():
    print("  %-30s %.3e %s  us/step %.3f  frac %s" % (k, v["value"], v["unit"], v["us_per_step"], v["roofline"]["frac"]))
print("cpu", d.get("cpu_baseline"))
PY
python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_steps20.json; cut -c1-400 $O/bench_steps20.json
python bench.py --mode fused --no-configs --no-cpu-baseline > $O/bench_fused.json; cut -c1-300 $O/bench_fused.json
python bench.py --mode step --no-configs --no-cpu-baseline > $O/bench_step.json; cut -c1-300 $O/bench_step.json
