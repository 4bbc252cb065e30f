#!/usr/bin/env python3
"""Turn what scripts/profile_round.sh left under gpurun_out/final/ into the tracked files under
profiles/<round>/ and profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    python scripts/profile_collect.py r01
"""
import contextlib
import io
import json
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


def counter_mean(db, kernel, counter):
    q = ("select dispatch_id, sum(value) from counters_collection where kernel_name like ? and counter_name = ? "
         "group by dispatch_id")
    v = [r[1] for r in sqlite3.connect(db).execute(q, (f"%{kernel}%", counter))]
    return sum(v) / len(v), len(v)


def greedy(dst):
    src = os.path.join(ROOT, "gpurun_out", "c5")
    for a, b in (("d2", "c5_greedy_depth2_65536"), ("d1", "c5_greedy_depth1_65536"), ("d2_1m", "c5_greedy_depth2_1048576"),
                 ("policy", "c5_greedy_policy_step_65536")):
        shutil.copy(os.path.join(src, a + ".json"), os.path.join(dst, b + ".json"))
    open(os.path.join(dst, "c5_greedy_kernel_stats.csv"), "w").write(
        capture(rocpd_summary.stats, os.path.join(src, "stats", "p_results.db")))
    sq = ("# k_greedy<4> depth 2 (pairs pooled over a tile, 4 wavefronts per tile, cheap + exact reply evaluation, twin "
          "placements evaluated once), 65536 boards = 1024 tiles; rocprofv3 --pmc, summed over instances per dispatch\n")
    sq += capture(rocpd_summary.counters, "k_greedy", [os.path.join(src, "pmc1", "p_results.db"),
                                                       os.path.join(src, "pmc2", "p_results.db")])
    open(os.path.join(dst, "c5_greedy_sq_counters.csv"), "w").write(sq)
    print(sq)
    print(open(os.path.join(dst, "c5_greedy_kernel_stats.csv")).read()[:400])


def main():
    rnd = sys.argv[1]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[2] == "greedy":
        return greedy(dst)
    for name in ("bench_default", "bench_maskonly", "bench_stepmode", "bench_4194304_boards", "bench_2097152_boards"):
        shutil.copy(os.path.join(SRC, name + ".json"), os.path.join(dst, "final_" + name + ".json"))
    for name in ("bench_c3_262144_boards", "bench_c2_4096_boards", "playouts"):
        shutil.copy(os.path.join(SRC, name + ".json"), os.path.join(dst, name + ".json"))
    for mode, kernel in (("fused", "k_rollout<true, true, 1, false>"), ("step", "k_step<true, true, 1>")):
        out = "final_%s_kernel_stats.csv" % ("fused" if mode == "fused" else "stepmode")
        open(os.path.join(dst, out), "w").write(capture(rocpd_summary.stats, os.path.join(SRC, mode + "_stats", "p_results.db")))
    traffic = {}
    rows = ["kernel,counter,dispatches,mean_value_KB"]
    for mode, kernel in (("fused", "k_rollout<true, true, 1, false>"), ("step", "k_step<true, true, 1>")):
        kb = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            kb[c], n = counter_mean(os.path.join(SRC, f"{mode}_pmc_{c}", "p_results.db"), kernel, c)
            rows.append(f"{kernel.replace(',', ';')},{c},{n},{kb[c]:.3f}")
        traffic[f"{mode}:1048576"] = {
            # gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md): FETCH_SIZE is doubled
            "hbm_bytes_per_launch": (2 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024,
            "FETCH_SIZE_KB": kb["FETCH_SIZE"], "WRITE_SIZE_KB": kb["WRITE_SIZE"],
            "note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, {kernel}, 2^20 boards; FETCH_SIZE "
                    f"doubled per MI355X_MICROARCH.md; source profiles/{rnd}/final_pmc_summary.csv "
                    f"(scripts/profile_round.sh + scripts/profile_collect.py)"}
    open(os.path.join(dst, "final_pmc_summary.csv"), "w").write("\n".join(rows) + "\n")
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    sq = "# k_rollout<true, true, 1>, 2^20 boards, one ply per launch; rocprofv3 --pmc (two passes), per dispatch\n"
    sq += capture(rocpd_summary.counters, "k_rollout<true, true, 1, false>", [os.path.join(SRC, "fused_sq1", "p_results.db"),
                                                                     os.path.join(SRC, "fused_sq2", "p_results.db")])
    open(os.path.join(dst, "final_fused_sq_counters.csv"), "w").write(sq)
    print(open(os.path.join(dst, "final_pmc_summary.csv")).read())
    print(sq)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in traffic.items()}))


if __name__ == "__main__":
    main()
