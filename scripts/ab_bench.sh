#!/bin/bash
# A/B timing of differently built libraries of the same ABI (GOBBLET_HIP_LIB): runs bench.py on each
# library in turn, ROUNDS times, and prints µs per ply.  usage: scripts/ab_bench.sh ROUNDS "bench args" lib1.so lib2.so ...
rounds=$1; args=$2; shift 2
mkdir -p gpurun_out
for r in $(seq "$rounds"); do
  for lib in "$@"; do
    GOBBLET_HIP_LIB=$lib python bench.py $args --no-cpu-baseline > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    python - "$lib" "$r" <<PY
import json, sys
d = json.loads([x for x in open("gpurun_out/ab.log") if x.startswith("{")][-1])
print(f"round {sys.argv[2]} {sys.argv[1]}: {d['ms_per_step'] * 1000:.2f} us/ply")
PY
  done
done
