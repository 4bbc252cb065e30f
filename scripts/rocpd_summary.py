#!/usr/bin/env python3
"""Summaries of rocprofv3's sqlite output (`-d DIR -o NAME` -> DIR/NAME_results.db), in the shape of
the files under profiles/:

    python scripts/rocpd_summary.py stats    RESULTS.db  > kernel_stats.csv
    python scripts/rocpd_summary.py counters KERNEL_SUBSTRING RESULTS.db [RESULTS2.db ...] > counters.csv
    python scripts/rocpd_summary.py bygrid   KERNEL_SUBSTRING RESULTS.db [SKIP] > per_size_stats.csv
"""
import sqlite3
import statistics
import sys


def stats(db):
    d = {}
    for name, dur in sqlite3.connect(db).execute("select name, duration from kernels"):
        d.setdefault(name, []).append(dur)
    tot = sum(sum(v) for v in d.values())
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev")
    for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print('"%s",%d,%d,%.6f,%.2f,%d,%d,%.6f' % (name, len(v), sum(v), sum(v) / len(v), 100 * sum(v) / tot, min(v),
                                                  max(v), statistics.pstdev(v)))


def counters(kernel, dbs):
    print("counter,dispatches,mean_value")
    for db in dbs:
        d = {}
        q = ("select counter_name, dispatch_id, sum(value) from counters_collection where kernel_name like ? "
             "group by counter_name, dispatch_id")
        for name, _, v in sqlite3.connect(db).execute(q, (f"%{kernel}%",)):
            d.setdefault(name, []).append(v)
        for name, vs in sorted(d.items()):
            print(f"{name},{len(vs)},{sum(vs) / len(vs):.1f}")


def bygrid(kernel, db, skip=0):
    """Durations of one kernel per (template instance, grid size): a sweep over batch sizes in one process.
    The first `skip` dispatches of every group (warm-up) are left out.  gap = start - previous kernel's end."""
    rows = list(sqlite3.connect(db).execute(
        "select name, grid_x, workgroup_x, start, end, duration from kernels order by start"))
    d, prev_end = {}, None
    for name, gx, wx, st, en, dur in rows:
        if kernel in name:
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            key = (short, gx // wx, wx)
            d.setdefault(key, []).append((dur, st - prev_end if prev_end else 0))
        prev_end = en
    print("kernel,workgroups,workgroup_size,dispatches,mean_ns,median_ns,min_ns,max_ns,median_gap_before_ns")
    for (name, wgs, wx), v in sorted(d.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        v = v[skip:] if len(v) > skip else v
        du = [x[0] for x in v]
        print('"%s",%d,%d,%d,%.1f,%.1f,%d,%d,%.1f' % (name, wgs, wx, len(du), sum(du) / len(du), statistics.median(du),
                                                    min(du), max(du), statistics.median(x[1] for x in v)))


if __name__ == "__main__":
    if sys.argv[1] == "bygrid":
        bygrid(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        counters(sys.argv[2], sys.argv[3:])
