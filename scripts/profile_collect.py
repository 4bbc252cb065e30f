#!/usr/bin/env python3
"""Turn what scripts/profile_round.sh left under gpurun_out/final/ into the tracked files under
profiles/<round>/ and profiles/pmc_traffic.json (read by bench.py for roofline.traffic and the greedy
VALU-instruction count; every record carries the hash of the kernel sources it was measured on).

    python scripts/profile_collect.py r02
"""
import contextlib
import hashlib
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


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("gobblet_hip.hip", "gobblet_device.h", "gobblet_knobs.h"):
        h.update(open(os.path.join(ROOT, "gobblet-rl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def reduced(run):
    """rows of gpurun_out/final/RUN.counters.csv (scripts/profile_reduce.py): [(kernel, counter, dispatches, mean, mean w/o first)]"""
    import csv
    path = os.path.join(SRC, run + ".counters.csv")
    if not os.path.exists(path):
        return []
    return [(r["kernel"], r["counter"], int(r["dispatches"]), float(r["mean_value"]), float(r["mean_without_first_dispatch"]))
            for r in csv.DictReader(open(path))]


def counter_mean(run, kernel, counter):
    """mean per dispatch of `counter` over the dispatches of the kernel whose name contains `kernel`, first dispatch of the
    process left out (one-off effects); (value, dispatches counted)"""
    for k, c, n, _, rest in reduced(run):
        if kernel in k and c == counter:
            return rest, max(1, n - 1)
    raise KeyError((run, kernel, counter))


def counters_table(kernel, runs):
    out = "counter,dispatches,mean_value\n"
    for run in runs:
        for k, c, n, mean, _ in sorted(reduced(run), key=lambda r: r[1]):
            if kernel in k:
                out += f"{c},{n},{mean:.1f}\n"
    return out


def main():
    rnd = sys.argv[1]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    khash = kernel_source_hash()
    for name in ("bench_default", "bench_driver_cmd", "bench_single_ply", "bench_stepmode", "bench_step_two_launch", "bench_maskonly",
                 "bench_c4_shard_131072", "greedy_65536", "greedy_1048576", "greedy_policy", "playouts"):
        if not os.path.exists(os.path.join(SRC, name + ".json")):  # (a trimmed round: scripts/profile_round.sh r5a / r5b)
            continue
        shutil.copy(os.path.join(SRC, name + ".json"), os.path.join(dst, name + ".json"))
        if os.path.exists(os.path.join(SRC, name + "_full.json")):  # (bench.py: the full record behind the compact line)
            shutil.copy(os.path.join(SRC, name + "_full.json"), os.path.join(dst, name + "_full.json"))
    for name in ("facade.txt", "valu_rates.txt", "winner_lanes.txt", "write_classes.txt", "placement_probe_check.txt",
                 "placement_ab.txt", "soak_parity.txt", "wave_placement.txt", "valu_mix.txt", "reply_rate.txt",
                 "greedy_wave_stamps.txt", "icache_cold.txt", "flag_sync.txt", "policy_wave_stamps.txt", "role_phase_stamps.txt",
                 "greedy_floor.txt", "quad_split.txt"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(dst, name))
    # instruction-cache counters of k_greedy at 65 536 and 2^20 boards: the kernel's rows only
    rows_ic = []
    for f in ("greedy_icache.counters.csv", "greedy_icache_1m.counters.csv", "icache_1048576.counters.csv"):
        if os.path.exists(os.path.join(SRC, f)):
            rows_ic += [ln for ln in open(os.path.join(SRC, f)).read().splitlines() if ln.startswith('"k_greedy') and ln not in rows_ic]
    if rows_ic:
        open(os.path.join(dst, "greedy_icache.counters.csv"), "w").write(
            "# rocprofv3 --pmc, gbl_greedy depth 2 (scripts/run_eager.py greedy): 65 536 boards (k_greedy<4, 16>) and 2^20 boards (k_greedy<1, 4>)\n"
            "kernel,counter,dispatches,mean_value,mean_without_first_dispatch\n" + "\n".join(rows_ic) + "\n")
    for run in ("collect", "single", "step", "greedy", "policy", "driver"):
        src = os.path.join(SRC, f"{run}_stats.kernel_stats.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, f"{run}_kernel_stats.csv"))
    if os.path.exists(os.path.join(SRC, "sweep_trace.by_grid.csv")):
        open(os.path.join(dst, "kernel_durations_by_size.csv"), "w").write(
            "# scripts/sweep_sizes.py under rocprofv3 --kernel-trace: k_rollout = one ply per launch, k_collect = 32 plies per launch\n"
            + open(os.path.join(SRC, "sweep_trace.by_grid.csv")).read())
    # ---- HBM traffic per launch ---------------------------------------------------------------------------------
    traffic = {}
    rows = ["kernel,counter,dispatches,mean_value_KB"]
    # run name (scripts/profile_round.sh: pmc_runs.txt) -> (key bench.py looks up, kernel name substring)
    runs = {"collect_T8": ("collect:1048576:T8", "k_collect<true, true"), "collect_T20": ("collect:1048576:T20", "k_collect<true, true"),
            "collect_4096_T1024": ("collect:4096:T1024", "k_collect5<true"),
            "collect_16384_T512": ("collect:16384:T512", "k_collect3<true, true"),
            "collect_32768_T256": ("collect:32768:T256", "k_collect3<true, true"),
            "collect_65536_T256": ("collect:65536:T256", "k_collect2<true, true"),
            "collect_4194304_T4": ("collect:4194304:T4", "k_collect<true, true"),
            "collect_131072_T128": ("collect:131072:T128", "k_collect2<true, true"),
            "collect_131072_T20": ("collect:131072:T20", "k_collect2<true, true"),
            "collect_262144_T64": ("collect:262144:T64", "k_collect<true, true"),
            "collect_262144_T20": ("collect:262144:T20", "k_collect<true, true"),
            "collect_524288_T32": ("collect:524288:T32", "k_collect<true, true"),
            "collect_524288_T20": ("collect:524288:T20", "k_collect<true, true"),
            "collect_4194304_T8": ("collect:4194304:T8", "k_collect<true, true"),
            "collect_noobs_T16": ("collect-noobs:1048576:T16", "k_collect3<true, false"),
            "fused_1048576": ("fused:1048576", "k_rollout<true, true"), "fused_262144": ("fused:262144", "k_rollout<true, true"),
            "fused_131072": ("fused:131072", "k_rollout<true, true"), "fused_4096": ("fused:4096", "k_rollout<true, true"),
            "fused_4194304": ("fused:4194304", "k_rollout<true, true"),
            "fused_noobs_1048576": ("fused-noobs:1048576", "k_rollout<true, false"),
            "step_1048576": ("step:1048576", "k_step<true, true"), "step_131072": ("step:131072", "k_step<true, true"),
            "step2_1048576": ("step2:1048576", "k_step<true, true")}
    for run, (key, kernel) in runs.items():
        if not os.path.exists(os.path.join(SRC, f"pmc_{run}_FETCH_SIZE.counters.csv")):
            continue
        kb = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            kb[c], n = counter_mean(f"pmc_{run}_{c}", kernel, c)
            rows.append(f"{kernel.replace(',', ';')} [{key}],{c},{n},{kb[c]:.3f}")
        traffic[key] = {
            # gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md): FETCH_SIZE is doubled
            "hbm_bytes_per_launch": (2 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024,
            "FETCH_SIZE_KB": kb["FETCH_SIZE"], "WRITE_SIZE_KB": kb["WRITE_SIZE"], "kernel_source_hash": khash,
            "source": f"profiles/{rnd}/pmc_summary.csv",
            "note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, eager launches of {kernel}...>, "
                    f"{key.split(':')[1]} boards; FETCH_SIZE doubled per MI355X_MICROARCH.md (scripts/profile_round.sh + profile_collect.py)"}
    open(os.path.join(dst, "pmc_summary.csv"), "w").write("\n".join(rows) + "\n")
    # ---- SQ counters ------------------------------------------------------------------------------------------------
    for m, kernel, what in (("traj", "k_collect<true, true", "gbl_collect FULL, 8 plies per launch, 2^20 boards"),
                            ("trajmask", "k_collect3<true, false", "gbl_collect MASK_ONLY (k_collect3: a playing and a mask-row wavefront per tile), 8 plies per launch, 2^20 boards"),
                            ("full", "k_rollout<true, true", "gbl_rollout FULL, one ply per launch, 2^20 boards"),
                            ("mask", "k_rollout<true, false", "gbl_rollout MASK_ONLY, one ply per launch, 2^20 boards"),
                            ("greedy", "k_greedy", "gbl_greedy depth 2, 65536 boards"),
                            ("policy", "k_collect_policy", "gbl_collect_policy greedy vs greedy, 16 plies per launch, 65536 boards")):
        sq = f"# {what}; rocprofv3 --pmc (two passes), summed over instances, mean per dispatch\n"
        if not reduced(f"{m}_sq1"):
            continue
        sq += counters_table(kernel, [f"{m}_sq1", f"{m}_sq2"])
        open(os.path.join(dst, f"sq_counters_{m}.csv"), "w").write(sq)
        if m == "greedy":
            insts, _ = counter_mean("greedy_sq1", "k_greedy", "SQ_INSTS_VALU")
            traffic["greedy:65536"] = {"SQ_INSTS_VALU": insts, "kernel_source_hash": khash,
                                       "source": f"profiles/{rnd}/sq_counters_greedy.csv"}
        if m == "policy":
            insts, _ = counter_mean("policy_sq1", "k_collect_policy", "SQ_INSTS_VALU")
            traffic["policy-collect:65536:T16"] = {"SQ_INSTS_VALU": insts, "kernel_source_hash": khash,
                                                   "source": f"profiles/{rnd}/sq_counters_policy.csv"}
        print(sq)
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "pmc_summary.csv")).read())
    if os.path.exists(os.path.join(dst, "kernel_durations_by_size.csv")):
        print(open(os.path.join(dst, "kernel_durations_by_size.csv")).read())
    print(json.dumps({k: v.get("hbm_bytes_per_launch", v.get("SQ_INSTS_VALU")) for k, v in traffic.items()}))


if __name__ == "__main__":
    main()
