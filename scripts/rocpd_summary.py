#!/usr/bin/env python3
"""Summaries of rocprofv3's sqlite output (`-d DIR -o NAME` -> DIR/NAME_results.db), in the shape of
the files under profiles/:

    python scripts/rocpd_summary.py stats    RESULTS.db  > kernel_stats.csv
    python scripts/rocpd_summary.py counters KERNEL_SUBSTRING RESULTS.db [RESULTS2.db ...] > counters.csv
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        counters(sys.argv[2], sys.argv[3:])
