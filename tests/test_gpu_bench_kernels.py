"""GPU parity tests (-m gpu): every kernel instantiation that ``bench.py`` times, compared DIRECTLY with the CPU oracle
at the batch size (and through the dispatch branch) the benchmark record uses -- not transitively through another
kernel.  One test case per ``configs`` key of bench.py's JSON line (and the headline); the table below names the
instantiation each one reaches.  Bit-exact (integer / byte work).

    bench record                  entry point (as bench.py calls it)        kernel instantiation
    ----------------------------  ----------------------------------------  -------------------------------------------
    headline (2^20, collect)      gbl_collect, ply index on the device      k_collect<mask, obs, DEV_PLY, NT>, 8 and 20 plies per launch
    c2_4096                       gbl_collect                               k_collect5<obs, DEV_PLY>  (FULL <= 9 216 boards, MASK_ONLY <= 32 768: 32-board groups, hand-over ring)
    c_16384, c_32768              gbl_collect                               k_collect3<mask, obs, DEV_PLY, HAND>  (9 217 ... 45 056 boards)
    c_65536                       gbl_collect                               k_collect2<mask, obs, DEV_PLY>
    c3_262144                     gbl_collect                               k_collect<mask, obs, DEV_PLY, NT>
    c4_shard_131072               gbl_collect                               k_collect2<mask, obs, DEV_PLY>  (up to 2560 tiles)
    large_4194304                 gbl_collect                               k_collect<mask, obs, DEV_PLY, NT>, identity tile map
    maskonly_1048576              gbl_collect, obs_traj = NULL              k_collect3<mask, -, DEV_PLY, no HAND>  (MASK_ONLY up to 3 * 2^20 boards)
    single_ply_{1048576,262144,   gbl_rollout_at(plies = 1)                 k_rollout<mask, obs, NT = 1, DEV_PLY, ONE_PLY>
      131072,4096}
    single_ply_large_4194304      gbl_rollout_at(plies = 1)                 k_rollout<mask, obs, NT = 3, DEV_PLY, ONE_PLY>
    single_ply_maskonly_1048576   gbl_rollout_at(plies = 1), obs_out = NULL k_rollout<mask, -, NT = 1, DEV_PLY, ONE_PLY>
    step_two_launch_1048576       gbl_sample_at + gbl_step (--mode step2)   k_sample, k_step<mask, obs, NT, false>
    step_pipeline_{1048576,       gbl_step_ex, next_actions = actions       k_step<mask, obs, NT, EXT>: the next mover's draw fused
      131072}                       (--mode step)                             into the step's launch
    c5_greedy_65536               gbl_greedy                                k_greedy<4>  (test_gpu_parity.py::test_greedy_config5_full_size)
    greedy_collect_65536          gbl_collect_policy                        k_collect_policy<W> (test_gpu_policy_collect.py)
The oracle functions follow gobblet.py:179-271 (observe / step / reset) and board.py:82-220; see oracle/gobblet_oracle.c.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
THREADS = 16


@pytest.fixture(scope="module")
def G():
    import gobblet_rl_amd as g
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    g._native.lib()
    return g


def npy(x):
    return x.cpu().numpy()


def warm_pair(G, n, seed, base, warm, **kw):
    """An environment and the oracle's arrays after `warm` masked-random plies from reset (the benchmark's workload)."""
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed, env_base=base, **kw)
    s, tm, dn = oracle.batch_reset(n)
    if warm:
        env.rollout(warm)
        oracle.batch_rollout(s, tm, dn, seed, base, 0, warm, threads=THREADS, want_obs=False, want_mask=False)
    return env, s, tm, dn


def slot(tr, key, t_, n):
    """Ply t_ of trajectory entry `key` as an (n, ...) tensor, whichever the layout."""
    if tr["_layout"] == "time":
        return tr[key][t_]
    v = tr[key][:, t_]
    return v.reshape((v.shape[0] * 64,) + tuple(v.shape[2:]))[:n]


def check_trajectory(env, tr, T, ply0, s, tm, dn, illegal, with_obs=True, with_mask=True):
    """Every slot of a collected trajectory against the oracle's fused ply (sample -> step -> auto-reset -> observe)."""
    n, ended = env.num_envs, 0
    for t_ in range(T):
        o = oracle.batch_rollout(s, tm, dn, env.seed, env.env_base, ply0 + t_, 1, illegal_mode=illegal, threads=THREADS,
                                 want_obs=with_obs)
        assert np.array_equal(npy(slot(tr, "actions", t_, n)), o["actions"]), ("actions", t_)
        assert np.array_equal(npy(slot(tr, "winner", t_, n)), o["winner"]), ("winner", t_)
        assert np.array_equal(npy(slot(tr, "rewards", t_, n)), o["reward"]), ("rewards", t_)
        assert np.array_equal(npy(slot(tr, "done", t_, n)), dn), ("done", t_)
        assert np.array_equal(npy(slot(tr, "to_move", t_, n)), tm), ("to_move", t_)
        if with_mask:
            assert np.array_equal(npy(slot(tr, "action_mask", t_, n)), o["mask"]), ("action_mask", t_)
        if with_obs:
            assert np.array_equal(npy(slot(tr, "observation", t_, n)), o["obs"]), ("observation", t_)
        ended += int(dn.sum())
    assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm) and np.array_equal(npy(env.done), dn)
    return ended


# bench record -> (boards, plies per launch as bench.py's auto_traj picks them (cut where the oracle needs the time), obs)
COLLECT_RECORDS = {
    "headline_1048576": (1 << 20, 8, True),
    "headline_driver_cmd_1048576": (1 << 20, 20, True),   # `--steps 20`: the whole timed run is ONE launch of 20 plies
    "c2_4096": (4096, 1024, True),                         # (the long launches of the small shards: the record's own length)
    "c_16384": (16384, 128, True),
    "c_32768": (32768, 64, True),
    "c_65536": (65536, 48, True),
    "c3_262144": (262144, 16, True),
    "c4_shard_131072": (131072, 32, True),
    "large_4194304": (1 << 22, 3, True),
    "maskonly_1048576": (1 << 20, 16, False),
}


@pytest.mark.parametrize("record", list(COLLECT_RECORDS))
def test_bench_collect_record_vs_oracle(G, record):
    """gbl_collect exactly as bench.py's Pipeline launches it for this record (ply index in device memory, time-major
    slots, the record's batch size and so its dispatch branch), every slot against the oracle."""
    n, T, with_obs = COLLECT_RECORDS[record]
    seed, base, warm = 0, 0, 12
    env, s, tm, dn = warm_pair(G, n, seed, base, warm, with_observation=with_obs)
    if record != "headline_driver_cmd_1048576":       # (ONE eager launch: bench.py passes the ply index by value there)
        env.device_ply()                              # DEV_PLY instantiation: what the hipGraph replay needs
    tr = env.trajectory_buffers(T, placement="any")
    env.collect(T, out=tr, refresh=False)
    env.advance_ply()
    torch.cuda.synchronize()
    ended = check_trajectory(env, tr, T, warm, s, tm, dn, 0, with_obs=with_obs)
    assert ended > 0                                  # games ended (and restarted) inside the trajectory
    assert env.ply == warm + T


@pytest.mark.parametrize("n,with_obs,layout,illegal,device_ply", [
    # one board more than the two-wavefronts-per-tile kernel takes (2 560 tiles): the first grid of the one-wavefront k_collect
    (163841, True, "time", "noop", False), (163841, False, "time", "terminate", True),
    (163841, True, "tile", "terminate", True), (163841, False, "tile", "noop", False),
    # 2^20 boards x 4 plies, MASK_ONLY and FULL, both illegal modes
    (1 << 20, False, "time", "terminate", False), (1 << 20, True, "tile", "noop", False),
    # MASK_ONLY stays with k_collect3 up to 3 * 2^20 boards (the cases above): the first grid of k_collect<mask>, ragged
    (3 * (1 << 20) + 1, False, "time", "noop", True)])
def test_collect_one_wavefront_kernel_vs_oracle(G, n, with_obs, layout, illegal, device_ply):
    """The large grids of gbl_collect directly against the oracle -- FULL: k_collect (one wavefront per tile: grids above 2 048
    tiles); MASK_ONLY: k_collect3's playing + mask-row wavefronts up to 3 * 2^20 boards, k_collect beyond -- both layouts, both
    illegal modes, ply index by value and on the device; ragged last tiles at 163 841 and 3 145 729 boards."""
    variant = G._native.lib().gbl_collect_variant(n, 4, 1, int(with_obs))
    assert variant == (0 if with_obs or n > 3 * (1 << 20) else 4)   # (4 = GBL_COLLECT_TRIO)
    T, seed, base, warm = 4, 17, 5_000_000_000, 9
    env, s, tm, dn = warm_pair(G, n, seed, base, warm, with_observation=with_obs, illegal_mode=illegal)
    if device_ply:
        env.device_ply()
    tr = env.trajectory_buffers(T, layout=layout, placement="any")
    env.collect(T, out=tr, refresh=False)
    env.advance_ply()
    torch.cuda.synchronize()
    check_trajectory(env, tr, T, warm, s, tm, dn, 0 if illegal == "noop" else 1, with_obs=with_obs)


@pytest.mark.parametrize("n", [33, 3000, 163841])
@pytest.mark.parametrize("null", ["mask", "obs", "both", "scalars"])
def test_collect_c_abi_null_outputs(G, n, null):
    """gbl_collect called through the C-ABI with mask_traj / obs_traj / both / every scalar array NULL (k_collect and
    k_collect2 <false, *> and <*, false>; at 33 and 3 000 boards: the role kernel k_collect_small, which serves the launches WITHOUT a
    mask trajectory since round 6, with ragged groups, and k_collect5 with its scalar arrays NULL): what IS written equals the
    oracle, the state after the launch too."""
    nat, L = G._native, G._native.lib()
    T, seed, base, warm = 5, 23, 77, 7
    env, s, tm, dn = warm_pair(G, n, seed, base, warm)
    slot_boards = -(-n // 128) * 128
    dev = torch.device(DEV)
    z = lambda *shape, dtype=torch.int8: torch.zeros(shape, dtype=dtype, device=dev)  # noqa: E731
    act, win, rew = z(T, slot_boards, dtype=torch.int32), z(T, slot_boards), z(T, slot_boards, 2)
    don, tmv = z(T, slot_boards), z(T, slot_boards)
    mask, obs = z(T, slot_boards, 54), z(T, slot_boards, 3, 3, 13)
    want_mask, want_obs, want_scalars = null not in ("mask", "both"), null not in ("obs", "both"), null != "scalars"
    p = lambda x, on: x.data_ptr() if on else None  # noqa: E731
    nat.check(L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), p(act, want_scalars),
                            p(win, want_scalars), p(rew, want_scalars), p(don, want_scalars), p(tmv, want_scalars),
                            p(mask, want_mask), p(obs, want_obs), n, slot_boards, 64, seed, base, warm, None, T, 0, None, None,
                            nat.current_stream(dev)), "gbl_collect")
    torch.cuda.synchronize()
    for t_ in range(T):
        o = oracle.batch_rollout(s, tm, dn, seed, base, warm + t_, 1, threads=THREADS)
        if want_scalars:
            assert np.array_equal(npy(act[t_, :n]), o["actions"]) and np.array_equal(npy(win[t_, :n]), o["winner"])
            assert np.array_equal(npy(rew[t_, :n]), o["reward"]) and np.array_equal(npy(don[t_, :n]), dn)
            assert np.array_equal(npy(tmv[t_, :n]), tm)
        if want_mask:
            assert np.array_equal(npy(mask[t_, :n]), o["mask"]), t_
        if want_obs:
            assert np.array_equal(npy(obs[t_, :n]), o["obs"]), t_
    assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm)
    # the arrays handed over as NULL were not touched (they are separate allocations: nothing else could have been)
    if not want_mask:
        assert int(mask.abs().sum()) == 0
    if not want_obs:
        assert int(obs.abs().sum()) == 0
    if not want_scalars:
        assert int(act.abs().sum()) == 0 and int(don.abs().sum()) == 0


# bench record -> (boards, observation?)
SINGLE_PLY_RECORDS = {
    "single_ply_1048576": (1 << 20, True), "single_ply_262144": (262144, True), "single_ply_131072": (131072, True),
    "single_ply_4096": (4096, True), "single_ply_maskonly_1048576": (1 << 20, False),
    "single_ply_large_4194304": (1 << 22, True),
}


@pytest.mark.parametrize("record", list(SINGLE_PLY_RECORDS))
def test_bench_single_ply_record_vs_oracle(G, record):
    """gbl_rollout_at(plies = 1) -- the fused sample + step ply of bench.py's `fused` mode, ply index on the device --
    at the record's batch size (so with its non-temporal store policy: the mask stream too from 2^21 boards), three
    consecutive plies, every output against the oracle.  MASK_ONLY at 2^20 boards included."""
    n, with_obs = SINGLE_PLY_RECORDS[record]
    seed, base, warm = 3, 1 << 33, 10
    env, s, tm, dn = warm_pair(G, n, seed, base, warm, with_observation=with_obs)
    env.device_ply()
    for k in range(3):
        obs, rew, done, win = env.rollout(1)
        env.advance_ply()
        o = oracle.batch_rollout(s, tm, dn, seed, base, warm + k, 1, threads=THREADS, want_obs=with_obs)
        assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm) and np.array_equal(npy(done), dn)
        assert np.array_equal(npy(env.actions), o["actions"]) and np.array_equal(npy(win), o["winner"])
        assert np.array_equal(npy(rew), o["reward"]) and np.array_equal(npy(obs["action_mask"]), o["mask"])
        if with_obs:
            assert np.array_equal(npy(obs["observation"]), o["obs"])
    assert env.ply == warm + 3


@pytest.mark.parametrize("n", [1 << 20, 131072])
def test_bench_step_mode_vs_oracle(G, n):
    """bench.py --mode step2 at 2^20 boards (records step_two_launch_*): gbl_sample_at (ply index on the device) + gbl_step with
    auto-reset, two plies, every output against the oracle (k_sample, k_step<mask, obs, NT = 1, false>); then --mode step (records
    step_pipeline_*): gbl_step_ex with the next mover's draw written over the action array, three plies; then MASK_ONLY."""
    seed, base, warm = 4, 99, 11
    for with_obs in (True, False):
        env, s, tm, dn = warm_pair(G, n, seed, base, warm, with_observation=with_obs)
        env.device_ply()
        for k in range(2):
            a = env.sample_actions()
            exp_a = oracle.batch_sample(oracle.batch_legal_mask(s, tm), seed, base, warm + k)
            assert np.array_equal(npy(a), exp_a)
            obs, rew, done, win = env.step(a)
            env.advance_ply()
            o = oracle.batch_step(s, tm, dn, exp_a, auto_reset=True, threads=THREADS, want_obs=with_obs)
            assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(done), dn) and np.array_equal(npy(win), o["winner"])
            assert np.array_equal(npy(rew), o["reward"]) and np.array_equal(npy(obs["action_mask"]), o["mask"])
            if with_obs:
                assert np.array_equal(npy(obs["observation"]), o["obs"])
        # ... and the ONE-launch-per-ply form bench.py times now: gbl_step_ex draws the next mover's action from the mask it stores
        # (next_actions aliasing actions: one array carries the masked-random game from launch to launch; k_step<mask, obs, NT, EXT>)
        acts = env.sample_actions().clone()
        for k in range(2, 5):
            exp_a = oracle.batch_sample(oracle.batch_legal_mask(s, tm), seed, base, warm + k)
            assert np.array_equal(npy(acts), exp_a), k
            obs, rew, done, win = env.step(acts, next_actions=acts)
            env.advance_ply()
            o = oracle.batch_step(s, tm, dn, exp_a, auto_reset=True, threads=THREADS, want_obs=with_obs)
            assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(done), dn) and np.array_equal(npy(win), o["winner"])
            assert np.array_equal(npy(rew), o["reward"]) and np.array_equal(npy(obs["action_mask"]), o["mask"])
            if with_obs:
                assert np.array_equal(npy(obs["observation"]), o["obs"])
        assert np.array_equal(npy(acts), oracle.batch_sample(oracle.batch_legal_mask(s, tm), seed, base, warm + 5))


@pytest.mark.parametrize("n,T,illegal,with_obs", [(131072, 2, "noop", True), (262144, 2, "terminate", True),
                                                  (4099, 3, "terminate", False), (163841, 2, "noop", True), (70, 1, "noop", True),
                                                  (12001, 2, "noop", True), (40001, 1, "terminate", True), (40001, 3, "noop", False),
                                                  (20001, 2, "terminate", True), (5001, 2, "terminate", True)])
def test_collect_from_external_first_ply_vs_oracle(G, n, T, illegal, with_obs):
    """gbl_collect_from (bench records step_reply_*): the first ply plays caller-supplied actions -- an external policy's,
    some of them illegal or out of range -- the rest are sampled; k_collect2 at 131 072 boards, k_collect beyond 163 840.  Three
    launches in a row (the policy: the library's sampler on the previous launch's last mask, with wild actions thrown
    in), every slot against the oracle: batch_step for the given actions, batch_rollout for the replies."""
    seed, base, warm = 6, 314159, 8
    im = 0 if illegal == "noop" else 1
    env, s, tm, dn = warm_pair(G, n, seed, base, warm, with_observation=with_obs, illegal_mode=illegal)
    env.refresh()
    tr = env.trajectory_buffers(T, placement="any")
    rng = np.random.default_rng(1)
    mask = env.action_mask
    for launch in range(3):
        ply0 = warm + launch * T
        acts = oracle.batch_sample(npy(mask), seed + 1, base, ply0)          # "the external policy"
        wild = rng.random(n) < 0.15
        acts = np.where(wild, rng.integers(-3, 60, n), acts).astype(np.int32)
        exp_status = oracle.batch_action_status(s, tm, dn, acts, auto_reset=True)   # (before the oracle steps)
        status = torch.full((n,), 77, dtype=torch.int8, device=DEV) if launch else None
        env.collect(T, out=tr, first_actions=torch.from_numpy(acts).to(DEV), first_status=status, refresh=False)
        torch.cuda.synchronize()
        if status is not None:   # gbl_collect_from_ex: bit 0 = illegal, bit 1 = outside [0, 54)
            assert np.array_equal(npy(status), exp_status) and (exp_status == 3).any() == (n > 60), ("status", launch)
        for t_ in range(T):
            if t_ == 0:
                o = oracle.batch_step(s, tm, dn, acts, illegal_mode=im, auto_reset=True, threads=THREADS, want_obs=with_obs)
                exp_a = acts
            else:
                o = oracle.batch_rollout(s, tm, dn, seed, base, ply0 + t_, 1, illegal_mode=im, threads=THREADS, want_obs=with_obs)
                exp_a = o["actions"]
            assert np.array_equal(npy(tr["actions"][t_]), exp_a), ("actions", launch, t_)
            assert np.array_equal(npy(tr["winner"][t_]), o["winner"]) and np.array_equal(npy(tr["rewards"][t_]), o["reward"])
            assert np.array_equal(npy(tr["done"][t_]), dn) and np.array_equal(npy(tr["to_move"][t_]), tm), (launch, t_)
            assert np.array_equal(npy(tr["action_mask"][t_]), o["mask"]), ("mask", launch, t_)
            if with_obs:
                assert np.array_equal(npy(tr["observation"][t_]), o["obs"]), ("obs", launch, t_)
        assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm)
        mask = tr["action_mask"][T - 1]
    assert env.ply == warm + 3 * T
    with pytest.raises(ValueError):
        env.collect(T, out=tr, first_actions=torch.zeros(n, dtype=torch.int32, device=DEV), policies=("random", "random"))


# ---- batches that do not fill the chip: k_collect5 (round 6: 32-board groups, a playing wavefront + row wavefronts behind a hand-over
# ring; until then the role kernel k_collect_small, which still serves launches without a mask trajectory) and k_collect3 ----
# every form the dispatch can pick, with ragged last groups (1, 15, 17, 33, 63 rows: observation wavefronts whose share of the
# group is partial, or empty) and whole ones
SMALL_SIZES = [1, 15, 16, 17, 31, 33, 63, 65, 4096, 4099, 8192,   # k_collect5 (round 6): groups of 32 boards, a playing wavefront + row wavefronts
               8193, 8209, 8241, 9216,                           #   ... FULL up to 9 216 boards (MASK_ONLY up to 32 768)
               9217, 9249, 12321, 16384]                         # k_collect3 (FULL; MASK_ONLY still k_collect5)
TRIO_SIZES = [16385, 16447, 32768, 32801, 45056, 57344]          # k_collect3 (FULL 9 217 ... 45 056 boards; MASK_ONLY 32 769 ... 3 * 2^20)


def expected_small_variant(n, with_obs):
    """gbl_collect_variant for a launch WITH a mask trajectory: 5 = GBL_COLLECT_GROUP32 (k_collect5), 4 = GBL_COLLECT_TRIO (k_collect3),
    2 = GBL_COLLECT_PAIR (k_collect2)."""
    if with_obs:
        return 5 if n <= 9216 else 4 if n <= 45056 else 2
    return 5 if n <= 32768 else 4


@pytest.mark.parametrize("with_obs", [True, False], ids=["full", "maskonly"])
@pytest.mark.parametrize("n", SMALL_SIZES)
def test_small_batch_collect_vs_oracle(G, n, with_obs):
    """gbl_collect on batches that do not fill the chip (k_collect5, GBL_COLLECT_GROUP32; k_collect3 beyond its range)
    directly against the oracle, FULL and MASK_ONLY, time- and tile-major slots, both illegal modes, ply index by value and on
    the device, tallies and turn counters; ragged last sub-tiles and whole ones of every form."""
    assert G._native.lib().gbl_collect_variant(n, 7, 1, int(with_obs)) == expected_small_variant(n, with_obs)
    small_batch_case(G, n, with_obs)


@pytest.mark.parametrize("with_obs", [True, False], ids=["full", "maskonly"])
@pytest.mark.parametrize("n", TRIO_SIZES)
def test_trio_collect_vs_oracle(G, n, with_obs):
    """gbl_collect where k_collect3 runs (one playing wavefront per tile hands every ply's position to a mask-row and an
    observation-row wavefront: GBL_COLLECT_TRIO), against the oracle as above; ragged last tiles (1 and 63 rows) and whole ones."""
    assert G._native.lib().gbl_collect_variant(n, 7, 1, int(with_obs)) == expected_small_variant(n, with_obs)
    small_batch_case(G, n, with_obs)


def small_batch_case(G, n, with_obs):
    T, seed, base, warm = (7 if n <= 8192 else 5), 29, 123_456_789_012, 6
    for layout, illegal, device_ply in (("time", "noop", True), ("tile", "terminate", False)):
        env, s, tm, dn = warm_pair(G, n, seed, base, warm, with_observation=with_obs, illegal_mode=illegal, track_turn=True)
        turn = npy(env.turn).copy()
        if device_ply:
            env.device_ply()
        tr = env.trajectory_buffers(T, layout=layout, placement="any")
        before = env.counters.clone()
        env.collect(T, out=tr, refresh=False, count=True)
        env.advance_ply()
        torch.cuda.synchronize()
        ended = check_trajectory(env, tr, T, warm, s, tm, dn, 0 if illegal == "noop" else 1, with_obs=with_obs)
        delta = npy(env.counters - before)
        assert delta[0] == n * T and delta[1] == ended and delta[2] + delta[3] == ended
        # raw_env.turn: plies since the board's last reset (the masked-random sampler never plays an illegal action)
        games_per_board = sum(npy(slot(tr, "done", t_, n)).astype(np.int64) for t_ in range(T))
        last_end = np.full(n, -1)
        for t_ in range(T):
            last_end = np.where(npy(slot(tr, "done", t_, n)) != 0, t_, last_end)
        exp_turn = np.where(games_per_board > 0, T - 1 - last_end, turn + T)
        assert np.array_equal(npy(env.turn), exp_turn)
        if layout == "time":  # the padding boards of a slot are never written
            pad = tr["_full"]["action_mask"][:, n:]
            assert pad.numel() == 0 or int(pad.abs().sum()) == 0


@pytest.mark.parametrize("n", [1, 63, 65, 4096, 4099])
def test_small_batch_one_ply_entry_points_vs_oracle(G, n):
    """The one-ply entry points on small batches (k_rollout / k_step: routing them to the sub-tile kernel gained nothing and was
    taken out again): gbl_rollout(plies = 1) and gbl_sample + gbl_step with auto-reset, FULL and MASK_ONLY, both illegal modes,
    with wild (illegal / out-of-range) actions thrown at gbl_step; every output against the oracle."""
    seed, base, warm = 31, 7, 9
    rng = np.random.default_rng(n)
    for with_obs, illegal in ((True, "noop"), (False, "terminate")):
        im = 0 if illegal == "noop" else 1
        env, s, tm, dn = warm_pair(G, n, seed, base, warm, with_observation=with_obs, illegal_mode=illegal, track_turn=True)
        for k in range(4):
            obs, rew, done, win = env.rollout(1)
            o = oracle.batch_rollout(s, tm, dn, seed, base, warm + k, 1, illegal_mode=im, threads=THREADS, want_obs=with_obs)
            assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm) and np.array_equal(npy(done), dn)
            assert np.array_equal(npy(env.actions), o["actions"]) and np.array_equal(npy(win), o["winner"])
            assert np.array_equal(npy(rew), o["reward"]) and np.array_equal(npy(obs["action_mask"]), o["mask"])
            if with_obs:
                assert np.array_equal(npy(obs["observation"]), o["obs"])
        for k in range(4):
            acts = oracle.batch_sample(oracle.batch_legal_mask(s, tm), seed, base, warm + 4 + k)
            assert np.array_equal(npy(env.sample_actions()), acts)
            wild = rng.random(n) < 0.2
            acts = np.where(wild, rng.integers(-3, 60, n), acts).astype(np.int32)
            obs, rew, done, win = env.step(torch.from_numpy(acts).to(DEV))
            o = oracle.batch_step(s, tm, dn, acts, illegal_mode=im, auto_reset=True, threads=THREADS, want_obs=with_obs)
            assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm) and np.array_equal(npy(done), dn)
            assert np.array_equal(npy(win), o["winner"]) and np.array_equal(npy(rew), o["reward"])
            assert np.array_equal(npy(obs["action_mask"]), o["mask"])
            if with_obs:
                assert np.array_equal(npy(obs["observation"]), o["obs"])


def test_bench_config_keys_are_all_covered(G):
    """Every sub-record bench.py emits has a test above (or the named one elsewhere) that compares its kernel with the
    oracle: the key lists are read from bench.py itself, so a new record without a test fails here."""
    import bench
    keys = set(bench.CONFIG_RECORDS) | set(bench.EXTRA_RECORDS)
    covered = set(COLLECT_RECORDS) | set(SINGLE_PLY_RECORDS) | {
        "headline_recipe_1048576",   # = headline_1048576 above: 2^20 boards, 8 plies per launch, ply index on the device
        "c2_4096_T32", "c4_shard_131072_T32",   # = c2_4096 / c4_shard_131072 above (same kernel instantiations), 32 plies per launch
        "step_pipeline_1048576", "step_pipeline_131072", "step_two_launch_1048576",   # test_bench_step_mode_vs_oracle
        "c5_greedy_65536",         # test_gpu_parity.py::test_greedy_config5_full_size
        "greedy_collect_65536",    # test_gpu_policy_collect.py::test_policy_collect_config5_size_selfplay
        # gbl_collect_from, T = 2: test_collect_from_external_first_ply_vs_oracle
        "step_reply_131072", "step_reply_262144"}
    assert keys <= covered, keys - covered
