#!/bin/bash
# A/B sweep of library variants (scripts/build_variant.sh) over batch sizes, one GPU call:
#   scripts/experiments/ab_sweep.sh OUTDIR "SIZES" "MODES" lib1.so lib2.so ...      (a leading "check:" on a lib also runs the GPU parity tests on it)
# extra arguments for scripts/sweep_sizes.py in $SWEEP_ARGS
set -e -o pipefail
export TMPDIR=/tmp
O=$1; sizes=$2; modes=$3; shift 3
mkdir -p $O
: > $O/sweep.jsonl
for spec in "$@"; do
  lib=${spec#check:}
  if [ "$spec" != "$lib" ]; then
    GOBBLET_HIP_LIB=$lib timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/check_$(basename $lib .so).log 2>&1 || { tail -30 $O/check_$(basename $lib .so).log; exit 1; }
    tail -1 $O/check_$(basename $lib .so).log
  fi
  GOBBLET_HIP_LIB=$lib python scripts/sweep_sizes.py --sizes $sizes --modes $modes --plies 320 --reps 5 $SWEEP_ARGS >> $O/sweep.jsonl
done
python - $O/sweep.jsonl <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1])]
tags = []; sizes = []
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
    k = (r["mode"], r["boards"])
    if k not in sizes: sizes.append(k)
t = {(r["tag"], r["mode"], r["boards"]): r["us_per_ply"] for r in rows}
print("%-28s" % "lib" + "".join("%16s" % f"{m}:{b}" for m, b in sizes))
for tag in tags:
    print("%-28s" % tag.split("/")[-1] + "".join("%16.3f" % t.get((tag, m, b), float("nan")) for m, b in sizes))
PY
