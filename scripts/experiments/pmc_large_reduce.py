#!/usr/bin/env python3
"""Runs on the GPU box at the end of scripts/experiments/pmc_large.sh: every rocprofv3 database DIR/p_results.db under the given directory ->
DIR.counters.csv (per kernel and counter: dispatches, mean per dispatch of the sum over the counter's instances, and -- for
counters collected per instance -- the mean per dispatch of the smallest and the largest instance), databases deleted."""
import os
import shutil
import sqlite3
import sys

src = sys.argv[1]


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


for name in sorted(os.listdir(src)):
    d = os.path.join(src, name)
    db = os.path.join(d, "p_results.db")
    if not os.path.isdir(d) or not os.path.exists(db):
        continue
    con = sqlite3.connect(db)
    try:
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        rows = {}
        for kernel, counter, disp, v in con.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection"):
            rows.setdefault((short(kernel), counter), {}).setdefault(disp, []).append(v)
        out = ["# columns of counters_collection: " + " ".join(cols),
               "kernel,counter,dispatches,instances,mean_sum_per_dispatch,mean_min_instance,mean_max_instance"]
        for (kernel, counter), per in sorted(rows.items()):
            disp = sorted(per)[1:] if len(per) > 1 else sorted(per)  # (without the first dispatch)
            n = len(disp)
            out.append('"%s",%s,%d,%d,%.1f,%.1f,%.1f' % (kernel, counter, len(per), len(per[disp[0]]),
                                                          sum(sum(per[k]) for k in disp) / n, sum(min(per[k]) for k in disp) / n,
                                                          sum(max(per[k]) for k in disp) / n))
        open(d + ".counters.csv", "w").write("\n".join(out) + "\n")
    except sqlite3.Error as e:
        open(d + ".counters.csv", "w").write("# %s\n" % e)
    con.close()
    shutil.rmtree(d)
print("reduced:", sorted(f for f in os.listdir(src) if f.endswith(".csv")))
