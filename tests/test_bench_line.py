"""CPU-side guard of bench.py's output contract: the LAST stdout line must be a compact JSON record the driver can parse from
the tail of stdout (round 3's 20 kB line came back `parsed: null`).  No GPU: the record is built from a stand-in of a run."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def fake_full(world=8, configs=16, blob=2000):
    args = types.SimpleNamespace(no_obs=False, boards_per_gpu=0, mode="collect", traj=20, dist_backend="nccl")
    pipe = types.SimpleNamespace()
    roof = {"bound": "hbm", "achieved": 7002.2, "peak": 8000.0, "unit": "GB/s", "frac": 0.875, "traffic": 3837975851.0,
            "algorithmic_bytes_survey": 234, "frac_on_survey_bytes": 1.1349, "accounting": "a" * 100,
            "traffic_over_algorithmic": 1.012, "kernel": "k_collect (20 plies per launch)<mask,obs>", "algorithmic_bytes_per_env_step": 180.85,
            "algorithmic_bytes_per_launch": 3792699392.0, "mean_launch_us": 541.6, "launches_timed": 1, "timing": "t" * 300,
            "traffic_source": "s" * 300, "note": "n" * blob}
    placements = [{"ratio": 0.8 + 0.01 * r, "probes": [1.0] * 19, "held_gib": 62.0, "cap_gib": 64.0, "spread": True, "ended": "e" * 200}
                  for r in range(world)]
    full = bench.contract_record(args, pipe, roof, 1 << 20, (1 << 20) // world, world, 20, 5, 6.0e-4, 5.6e-4, 1, False,
                                 [72.0 + r for r in range(world)], placements, True, [541.6, 539.9, 545.0, 540.2, 560.1])
    full["configs"] = {f"record_{i}": {"workload": "w" * blob, "value": 1.0, "roofline": {"frac": 0.5, "note": "x" * blob}} for i in range(configs)}
    for i, name in enumerate(bench.COMPACT_CONFIGS):   # the sub-records the line itself carries
        full["configs"][name] = {"workload": "w" * blob, "value": 1.23456789e10 + i, "us_per_step": 27.123456,
                                 "roofline": {"frac": 0.87654321, "note": "x" * blob}}
    full["scale_prediction"] = {str(n): {"boards_per_gpu": (1 << 20) // n, "span_us": 100.0, "kernel_us": 80.0, "value_predicted": 2.0e11,
                                         "efficiency_predicted": 0.71234, "efficiency_kernel_only": 0.9} for n in (2, 4, 8)}
    full["cpu_baseline"] = {"value": 8.7e6, "unit": "env-steps/s", "cores": 16, "kind": "port", "sample": "s" * 400, "value_1core": 5.8e5,
                            "greedy_depth2": {"decisions_per_s_1core": 1.0e4, "sample": "g" * blob}}
    return full


def test_compact_line_is_small_flat_and_complete():
    full = fake_full()
    assert len(json.dumps(full)) > 40000                       # (the full record is what round 3 printed)
    text = bench.compact_line(full, "gpurun_out/bench_configs.json")
    assert "\n" not in text and len(text.encode()) < bench.COMPACT_LIMIT == 4096
    d = json.loads(text)
    for k in CONTRACT + ("cpu_baseline",):
        assert k in d, k
    assert "configs" not in d and "detail" not in d
    # every BASELINE config's [value, us per step, frac] and BOTH byte accountings are in the parsed line
    cc = d["configs_compact"]
    for name in ("c2_4096", "c3_262144", "c4_shard_131072", "c5_greedy_65536", "single_ply_1048576", "step_pipeline_1048576"):
        v = cc[name]
        assert len(v) == 3 and abs(v[0] - 1.2346e10) < 1e7 and v[1] == 27.12 and v[2] == 0.877, (name, v)
    r = d["roofline"]
    assert r["algorithmic_bytes_survey"] == 234 and r["frac_on_survey_bytes"] == 1.1349 and r["algorithmic_bytes_per_env_step"] == 180.85
    assert isinstance(r["accounting"], str) and len(r["accounting"]) <= 128   # (the driver's parser cuts longer strings)
    assert d["scale_prediction"]["8"] == [2.0e11, 0.712]
    assert d["config"]["value_contract_span"] == d["config"]["value_own_span"] * 5.6 / 6.0 or \
        abs(d["config"]["value_contract_span"] / d["value"] - 1) < 1e-5
    for sub in ("config", "roofline", "cpu_baseline"):
        assert all(not isinstance(v, (list, dict)) for v in d[sub].values()), sub
    assert d["roofline"]["frac"] == 0.875 and d["roofline"]["traffic"] and d["roofline"]["traffic_over_algorithmic"] == 1.012
    assert d["cpu_baseline"]["value"] == 8.7e6 and d["cpu_baseline"]["cores"] == 16 and d["cpu_baseline"]["kind"] == "port"
    c = d["config"]
    assert c["configs_file"] == "gpurun_out/bench_configs.json" and c["configs_recorded"] == 16 + len(bench.COMPACT_CONFIGS)
    assert c["kernel_us_max"] == 79.0 and c["kernel_us_min"] == 72.0 and c["rccl_ranks"] == 8
    assert abs(c["placement_ratio_min"] - 0.8) < 1e-9 and abs(c["placement_ratio_max"] - 0.87) < 1e-9
    # value / ms_per_step come from the CONTRACT's span (barrier + synchronize on both sides); the ranks' own spans beside them
    assert abs(d["value"] - (1 << 20) * 20 / 6.0e-4) / d["value"] < 1e-12 and abs(d["ms_per_step"] - 6.0e-4 / 20 * 1e3) < 1e-12
    assert c["ms_per_step_own_span"] < d["ms_per_step"] and c["value_own_span"] > c["value_contract_span"]
    # the line verifies itself: where the traffic figure comes from, the spread of five passes, whether the arrays were placed
    assert d["roofline"]["traffic_source"].startswith("sss") and len(d["roofline"]["traffic_source"]) <= 140
    assert c["kernel_us_median_of_5"] == 541.6 and c["kernel_us_min_of_5"] == 539.9 and c["kernel_us_max_of_5"] == 560.1
    assert c["placement"] == "partly placed"


def test_unplaced_arrays_are_flagged():
    full = fake_full(world=1)
    assert json.loads(bench.compact_line(full, None))["config"]["placement"] == "placed"
    full["config"]["placement_ratio_max"] = 1.0
    args = types.SimpleNamespace(no_obs=False, boards_per_gpu=0, mode="collect", traj=20, dist_backend="nccl")
    rec = bench.contract_record(args, None, {}, 1 << 20, 1 << 20, 1, 20, 5, 6e-4, 5.6e-4, 1, False, [540.0],
                                [{"ratio": 1.0, "probes": [1.0] * 15}], False, [540.0] * 5)
    assert rec["config"]["placement"] == "unplaced" and rec["config"]["placement_ratio_min"] == 1.0


def test_compact_line_sheds_prose_before_it_grows_too_long():
    full = fake_full()
    full["roofline"]["timing"] = "t" * 3000                    # (prose that would push the line past the limit is dropped, numbers never)
    d = json.loads(bench.compact_line(full, None))
    assert "timing" not in d["roofline"] and d["roofline"]["frac"] == 0.875 and len(json.dumps(d).encode()) < 4096


def test_valu_roofline_uses_the_two_cycle_peak():
    r = bench.valu_roofline("no-such-key", "k", 1e-5, 1, "t")
    assert r["peak"] == 1024 * 2.4e9 / 2 and r["frac"] is None   # (no committed counter for that key: null, with the reason)
    assert bench.VALU_CYCLES_PEAK == 2.0


def test_more_ranks_than_gpus_fails_fast_with_a_clear_message():
    """`--gpus N` on a node with fewer than N GPUs (the driver's SCALE run pointed at a one-GPU box; here: none at all) must
    end at once with a message -- bare, and under a launcher, where every rank has to leave BEFORE the rendezvous."""
    import subprocess
    import time
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    t0 = time.time()
    bare = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                          capture_output=True, text=True, timeout=120, env=env)
    assert bare.returncode != 0 and "--gpus 8 but this node has" in bare.stderr and bare.stdout.strip() == ""
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                               "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
                               "--warmup", "5"], capture_output=True, text=True, timeout=120, env=env)
    assert launched.returncode != 0 and "2 rank(s) but this node has" in launched.stderr
    assert time.time() - t0 < 100
