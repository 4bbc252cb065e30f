#!/bin/bash
# One GPU call: build + smoke, the whole GPU test suite, the driver's exact bench command.
#   gpurun --timeout 1100 -- 'scripts/gpu_check.sh [outdir]'
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-r05_check}
rm -rf $O && mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1 || { tail -20 $O/build_smoke.log; exit 1; }
tail -1 $O/build_smoke.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > $O/gputests.log 2>&1 || { tail -60 $O/gputests.log; exit 1; }
tail -12 $O/gputests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --configs-out $O/bench_configs.json > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { tail -20 $O/bench_driver_cmd.err; exit 1; }
tail -c 4200 $O/bench_driver_cmd.json
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_configs.json"))
for k, v in d.get("configs", {}).items():
    print("  %-30s %.3e %s  us/step %.3f  frac %s" % (k, v["value"], v["unit"], v["us_per_step"], v["roofline"]["frac"]))
PY
