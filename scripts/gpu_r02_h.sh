#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02_h
rm -rf $O && mkdir -p $O
for T in 4 8 16 32 64; do
  python scripts/sweep_sizes.py --sizes 131072,262144,1048576,4194304 --modes traj,trajmask,full --plies 640 --reps 3 --traj $T >> $O/sweep.jsonl
done
python - $O/sweep.jsonl <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
tags = []; sizes = []
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
    k = (r["mode"], r["boards"])
    if k not in sizes: sizes.append(k)
t = {(r["tag"], r["mode"], r["boards"]): r["us_per_ply"] for r in rows}
print("%-16s" % "lib" + "".join("%17s" % f"{m}:{b}" for m, b in sizes))
for tag in tags:
    print("%-16s" % tag.split("/")[-1] + "".join("%17.3f" % t.get((tag, m, b), float("nan")) for m, b in sizes))
PY
