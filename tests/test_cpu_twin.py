"""The HOST flavour of the ABI (gbl_cpu_*, include/gobblet_cpu.h, csrc/gobblet_cpu.cpp -- the device header compiled for the host)
against the oracle and the reference-generated golden vectors: the GPU parity tests themselves, run with device "cpu" at sizes
the CPU suite can afford.  BASELINE config 1 ("1 env ... via gobblet_v1.env() on CPU ... no GPU") runs here as written:
1 000 reference games through env(device="cpu").  The host flavour never touches oracle/; the oracle is the checker."""
import os

import numpy as np
import pytest
import torch

import oracle
from tests import test_gpu_bench_kernels as K
from tests import test_gpu_parity as P
from tests import test_gpu_policy_collect as C


@pytest.fixture()
def G(monkeypatch):
    import gobblet_rl_amd as g
    for mod in (P, K, C):
        monkeypatch.setattr(mod, "DEV", "cpu")
    monkeypatch.setattr(K, "THREADS", 4)
    monkeypatch.setattr(C, "THREADS", 4)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)

    def t(a, dtype=None):  # (a COPY: on the host torch.from_numpy would alias the oracle's own arrays, which the tests step beside)
        x = torch.from_numpy(np.array(a, copy=True))
        return x if dtype is None else x.to(dtype)
    monkeypatch.setattr(P, "t", t)
    host = g._native.lib_for("cpu")  # builds with g++ if needed
    monkeypatch.setattr(g._native, "lib", lambda: host)  # (tests that go to the C-ABI directly: the host flavour's entry points)
    return g


@pytest.mark.parametrize("n", [1, 65, 408])
def test_board_functions_vs_golden(G, golden_dir, n):
    P.test_board_functions_vs_golden(G, golden_dir, n)


def test_upstream_kat_and_golden_games(G, golden_dir):
    P.test_upstream_kat(G, golden_dir)
    P.test_step_vs_golden_games(G, golden_dir)


@pytest.mark.parametrize("n,illegal,auto_reset,with_obs", [(300, "noop", False, True), (4099, "terminate", True, False),
                                                           (2048, "noop", True, True), (65, "terminate", False, True)])
def test_step_vs_oracle(G, n, illegal, auto_reset, with_obs):
    P.test_step_vs_oracle(G, n, illegal, auto_reset, with_obs)


def test_board_api_eval_sampler_rollout_vs_oracle(G, golden_dir):
    P.test_step_optional_outputs_null(G)
    P.test_board_api_vs_oracle(G)
    P.test_board_eval_vs_oracle(G, 65)
    P.test_sampler_and_rollout_vs_oracle(G)
    P.test_turn_counter(G, golden_dir)


def test_greedy_vs_golden(G, golden_dir):
    P.test_decode_obs_and_greedy_vs_golden(G, golden_dir)
    P.test_greedy_restricted_masks(G, golden_dir)
    P.test_greedy_depth3(G, golden_dir)
    P.test_greedy_on_terminal_roots(G, golden_dir)
    P.test_greedy_policy_class(G, golden_dir)


def test_config1_a_thousand_reference_games_on_the_cpu(G, golden_dir):
    """BASELINE config 1 as written: gobblet_v1.env() on CPU, no GPU -- the reference's 1 000 masked-random games, ply by ply."""
    P.test_c1_thousand_reference_games(G, golden_dir)   # (its environments are made with device=DEV: "cpu" here)
    P.test_aec_facade_on_gpu(G, golden_dir)


@pytest.mark.parametrize("n,with_obs,illegal", [(65, True, "noop"), (4099, False, "terminate")])
def test_collect_equals_ply_by_ply_rollout(G, n, with_obs, illegal):
    P.test_collect_equals_ply_by_ply_rollout(G, n, with_obs, illegal)


@pytest.mark.parametrize("n,with_obs", [(17, True), (4099, True), (8209, False)])
def test_collect_vs_oracle(G, n, with_obs):
    K.small_batch_case(G, n, with_obs)


def test_collect_from_external_first_ply(G):
    K.test_collect_from_external_first_ply_vs_oracle(G, 4099, 3, "terminate", False)
    K.test_collect_from_external_first_ply_vs_oracle(G, 70, 1, "noop", True)


@pytest.mark.parametrize("n,T,policies,opening,kw", [
    (65, 14, ("greedy", "random"), 0, {"layout": "tile"}),
    (1500, 10, ("greedy", "greedy"), 2, {"launches": 2, "device_ply": True}),
    (700, 8, ("greedy1", "greedy3"), 0, {"illegal": "terminate", "with_obs": False}),
])
def test_policy_collect_vs_oracle(G, n, T, policies, opening, kw):
    C.run_and_check(G, n, T, policies, opening, **kw)


def test_greedy_policy_step_and_selfplay(G):
    P.test_greedy_policy_step_vs_oracle(G)
