#!/usr/bin/env python3
"""Runs ON THE GPU BOX at the end of a scripts/profile_round.sh stage: turns every rocprofv3 database under
gpurun_out/final/ (DIR/p_results.db) into small text summaries next to it and deletes the database -- gpurun copies at
most 64 MiB back, and a round's databases are several times that.

    DIR.kernel_stats.csv     per kernel: calls, total / mean / min / max duration (rocpd_summary.stats)
    DIR.by_grid.csv          (kernel traces only) per kernel instance and grid size (rocpd_summary.bygrid)
    DIR.counters.csv         (--pmc runs) kernel, counter, dispatches, mean over all dispatches, mean without the first
"""
import contextlib
import io
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import rocpd_summary  # noqa: E402

SRC = os.path.join(ROOT, "gpurun_out", "final")


def capture(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def counters(db):
    rows = {}
    q = ("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection "
         "group by kernel_name, counter_name, dispatch_id order by dispatch_id")
    for kernel, counter, _, v in sqlite3.connect(db).execute(q):
        rows.setdefault((short(kernel), counter), []).append(v)
    out = ["kernel,counter,dispatches,mean_value,mean_without_first_dispatch"]
    for (kernel, counter), v in sorted(rows.items()):
        rest = v[1:] if len(v) > 1 else v
        out.append('"%s",%s,%d,%.3f,%.3f' % (kernel, counter, len(v), sum(v) / len(v), sum(rest) / len(rest)))
    return "\n".join(out) + "\n"


def main():
    for name in sorted(os.listdir(SRC)):
        d = os.path.join(SRC, name)
        db = os.path.join(d, "p_results.db")
        if not os.path.isdir(d) or not os.path.exists(db):
            continue
        con = sqlite3.connect(db)
        tables = {r[0] for r in con.execute("select name from sqlite_master where type in ('table', 'view')")}
        con.close()
        try:
            has_counters = "counters_collection" in tables and sqlite3.connect(db).execute(
                "select count(*) from counters_collection").fetchone()[0] > 0
        except sqlite3.Error:
            has_counters = False
        if has_counters:
            open(d + ".counters.csv", "w").write(counters(db))
        else:
            open(d + ".kernel_stats.csv", "w").write(capture(rocpd_summary.stats, db))
            if name == "sweep_trace":
                open(d + ".by_grid.csv", "w").write(
                    capture(rocpd_summary.bygrid, "k_rollout", db, 64) + capture(rocpd_summary.bygrid, "k_collect", db, 1))
        shutil.rmtree(d)
    print("reduced:", len([f for f in os.listdir(SRC) if f.endswith(".csv")]), "summaries;",
          sum(os.path.getsize(os.path.join(SRC, f)) for f in os.listdir(SRC) if os.path.isfile(os.path.join(SRC, f))) >> 10, "KiB left")


if __name__ == "__main__":
    main()
