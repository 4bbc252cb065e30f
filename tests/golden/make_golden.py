#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (the reference lives at /root/reference and
never travels to the GPU box).  It imports the reference's own
``gobblet_rl/game/board.py``, ``gobblet.py`` and ``greedy_policy.py`` and
records inputs + outputs as plain data (.npz / .json).  No reference source is
copied: the files written here hold arrays only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

``gobblet.py`` imports gymnasium / pygame / pettingzoo, which are not
installed here; SURVEY.md App. C: minimal in-memory stand-ins for those three
third-party packages are registered in ``sys.modules`` first, so that the
bodies of ``raw_env.observe/_legal_moves/step/reset`` that execute are the
reference's own code.  Only ``AECEnv._accumulate_rewards`` and
``agent_selector`` are stand-ins (restated from PettingZoo 1.22.3).
"""
import ast
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


# --------------------------------------------------------------------------------------
def install_standins():
    gymnasium = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    spaces.Discrete = type("Discrete", (_Space,), {})
    spaces.Box = type("Box", (_Space,), {})
    spaces.Dict = type("Dict", (_Space,), {})
    gymnasium.spaces = spaces
    gymnasium.logger = types.SimpleNamespace(warn=lambda *a, **k: None)
    pygame = types.ModuleType("pygame")
    pettingzoo = types.ModuleType("pettingzoo")

    class AECEnv:
        def __init__(self):
            pass

        def _accumulate_rewards(self):
            for agent, reward in self.rewards.items():
                self._cumulative_rewards[agent] += reward

        def close(self):
            pass

    pettingzoo.AECEnv = AECEnv
    utils = types.ModuleType("pettingzoo.utils")

    class agent_selector:
        def __init__(self, agent_order):
            self.reinit(agent_order)

        def reinit(self, agent_order):
            self.agent_order = agent_order
            self._current_agent = 0
            self.selected_agent = 0

        def reset(self):
            self.reinit(self.agent_order)
            return self.next()

        def next(self):
            self._current_agent = (self._current_agent + 1) % len(self.agent_order)
            self.selected_agent = self.agent_order[self._current_agent - 1]
            return self.selected_agent

    utils.agent_selector = agent_selector
    utils.wrappers = types.SimpleNamespace()
    conversions = types.ModuleType("pettingzoo.utils.conversions")
    conversions.parallel_wrapper_fn = lambda f: f
    utils.conversions = conversions
    pettingzoo.utils = utils
    for name, mod in [("gymnasium", gymnasium), ("gymnasium.spaces", spaces), ("pygame", pygame),
                      ("pettingzoo", pettingzoo), ("pettingzoo.utils", utils),
                      ("pettingzoo.utils.conversions", conversions)]:
        sys.modules[name] = mod


sys.path.insert(0, REF)
install_standins()
from gobblet_rl.game.board import Board  # noqa: E402
from gobblet_rl.game.gobblet import raw_env  # noqa: E402
from gobblet_rl.game.greedy_policy import GreedyGobbletPolicy  # noqa: E402


def i8(a):
    a = np.asarray(a)
    assert np.all(a == np.round(a))
    return a.astype(np.int8)


# --------------------------------------------------------------------------------------
def kat_collector():
    """Literal arrays of tests/test_manual_policy_collector.py (upstream known-answer test)."""
    src = open(os.path.join(REF, "tests/test_manual_policy_collector.py")).read()
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
            if name.startswith("output") and name != "output7":
                v = node.value
                if isinstance(v, ast.Call):  # np.array([...])
                    v = v.args[0]
                found[name] = ast.literal_eval(v)
    kat = {
        "source": "tests/test_manual_policy_collector.py:49-507",
        "actions": [18, 36, 28, 46],
        "illegal_action": 29,
        "mask_after": {k: [int(bool(x)) for x in np.array(found[k]).reshape(-1)]
                       for k in ["output0", "output1", "output2", "output3", "output4", "output5"]},
        "legal_moves_output6": [int(x) for x in found["output6"]],
        "board_output8": [int(x) for x in np.array(found["output8"]).reshape(-1)],
    }
    # replay through the reference and note which literals the reference itself reproduces
    env = raw_env()
    env.reset()
    got = {"output0": env.observe(env.agent_selection)["action_mask"]}
    for name, a in zip(["output1", "output2", "output3", "output4"], kat["actions"]):
        env.step(a)
        got[name] = env.observe(env.agent_selection)["action_mask"]
    got["output5"] = got["output4"]
    legal6 = env._legal_moves()
    env.step(kat["illegal_action"])
    board8 = i8(env.board.squares)
    kat["reference_reproduces"] = {k: bool(np.array_equal(got[k], np.array(kat["mask_after"][k], np.int8)))
                                   for k in kat["mask_after"]}
    kat["reference_reproduces"]["output6"] = bool(legal6 == kat["legal_moves_output6"])
    kat["reference_reproduces"]["output8"] = bool(np.array_equal(board8, np.array(kat["board_output8"], np.int8)))
    kat["reference_mask_after"] = {k: [int(x) for x in got[k]] for k in got}
    kat["reference_board_after_illegal"] = [int(x) for x in board8]
    kat["reference_to_move_after_illegal"] = env.agents.index(env.agent_selection)
    with open(os.path.join(OUT, "kat_collector.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("kat_collector.json", kat["reference_reproduces"])


# --------------------------------------------------------------------------------------
def random_games(n_games=96, seed=20240607, p_any=0.08):
    """Seeded games through the reference raw_env (masked-random, with a few
    arbitrary -- possibly illegal -- actions to pin the silent no-op)."""
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ["game", "ply", "mover", "action", "squares_before", "squares_after", "to_move_after",
                           "mask_next", "mask_offturn", "winner", "done", "obs_p1", "obs_p2", "reward",
                           "cum_reward", "legal_before"]}
    for g in range(n_games):
        env = raw_env()
        env.reset()
        ply = 0
        while True:
            mover = env.agents.index(env.agent_selection)
            mask = env.observe(env.agent_selection)["action_mask"]
            if rng.random() < p_any:
                a = int(rng.integers(0, 54))
            else:
                a = int(rng.choice(np.flatnonzero(mask)))
            before = i8(env.board.squares)
            env.step(a)
            nxt = env.agent_selection
            other = env.agents[1 - env.agents.index(nxt)]
            rec["game"].append(g); rec["ply"].append(ply); rec["mover"].append(mover); rec["action"].append(a)
            rec["legal_before"].append(int(mask[a]))
            rec["squares_before"].append(before)
            rec["squares_after"].append(i8(env.board.squares))
            rec["to_move_after"].append(env.agents.index(nxt))
            rec["mask_next"].append(env.observe(nxt)["action_mask"])
            rec["mask_offturn"].append(env.observe(other)["action_mask"])
            rec["winner"].append(env.board.check_for_winner())
            rec["done"].append(int(env.terminations["player_1"]))
            rec["obs_p1"].append(env.observe("player_1")["observation"])
            rec["obs_p2"].append(env.observe("player_2")["observation"])
            rec["reward"].append([env.rewards["player_1"], env.rewards["player_2"]])
            rec["cum_reward"].append([env._cumulative_rewards["player_1"], env._cumulative_rewards["player_2"]])
            ply += 1
            if env.terminations["player_1"] or ply >= 60:
                break
    out = {k: np.asarray(v) for k, v in rec.items()}
    for k in ["squares_before", "squares_after", "mask_next", "mask_offturn", "obs_p1", "obs_p2"]:
        out[k] = out[k].astype(np.int8)
    for k in ["mover", "to_move_after", "winner", "done", "reward", "cum_reward", "legal_before"]:
        out[k] = out[k].astype(np.int8)
    out["game"] = out["game"].astype(np.int32); out["ply"] = out["ply"].astype(np.int32)
    out["action"] = out["action"].astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "random_games.npz"), **out)
    n = len(out["action"])
    print(f"random_games.npz: {n_games} games, {n} plies, illegal plies {int((out['legal_before'] == 0).sum())}, "
          f"mover-loses terminals {int(((out['winner'] != 0) & (out['winner'] == np.where(out['mover'] == 0, -1, 1))).sum())}")
    return out


# --------------------------------------------------------------------------------------
def board_fn_vectors(squares_list):
    """Board-level functions of board.py on a list of 27-vectors."""
    rows = {k: [] for k in ["squares", "flatboard", "covered", "legal_p1", "legal_p2", "winner", "game_over",
                            "get_action_p1", "get_action_p2", "obs_p1", "obs_p2"]}
    env = raw_env()
    env.reset()
    for sq in squares_list:
        b = Board()
        b.squares = np.array(sq, dtype=np.float64)
        rows["squares"].append(i8(sq))
        rows["flatboard"].append(i8(b.get_flatboard()))
        rows["covered"].append(i8(b.check_covered()))
        rows["legal_p1"].append([int(b.is_legal(a, 0)) for a in range(54)])
        rows["legal_p2"].append([int(b.is_legal(a, 1)) for a in range(54)])
        rows["winner"].append(b.check_for_winner())
        rows["game_over"].append(int(b.check_game_over()))
        rows["get_action_p1"].append([[b.get_action(p, s, 0) for s in (1, 2, 3)] for p in range(9)])
        rows["get_action_p2"].append([[b.get_action(p, s, 1) for s in (1, 2, 3)] for p in range(9)])
        env.board = b
        rows["obs_p1"].append(env.observe("player_1")["observation"])
        rows["obs_p2"].append(env.observe("player_2")["observation"])
    return {k: np.asarray(v).astype(np.int8) for k, v in rows.items()}


def random_valid_board(rng):
    """A random board obeying the placement invariants (each piece at most once,
    on its own level, stacks strictly increasing) -- denser than self-play reaches."""
    sq = np.zeros(27, np.int8)
    for piece in range(1, 7):
        for sign in (1, -1):
            if rng.random() < 0.75:
                level = (piece - 1) // 2
                free = [p for p in range(9) if sq[9 * level + p] == 0]
                p = int(rng.choice(free))
                sq[9 * level + p] = sign * piece
    return sq


def edge_cases():
    """SURVEY.md App. D quirks as explicit boards."""
    E = {}

    def put(pairs):
        sq = np.zeros(27, np.int8)
        for cell, v in pairs:
            sq[cell] = v
        return sq

    # dual lines, both orders (last matching line decides)
    E["dual_p1_line0_p2_line2"] = put([(18 + 0, 5), (18 + 1, 6), (9 + 2, 3), (9 + 6, -3), (9 + 7, -4), (18 + 8, -5)])
    E["dual_p2_line0_p1_line2"] = -E["dual_p1_line0_p2_line2"]
    # same line index for both is impossible; diagonal vs column
    E["p1_diag_p2_col"] = put([(18 + 0, 5), (9 + 4, 3), (18 + 8, 6), (9 + 3, -3), (18 + 5, -5), (0 + 1, -1)])
    # covered small under medium / large, covered medium under large
    E["covered_stack"] = put([(0, 1), (9, -3), (18, 5), (1, -2), (18 + 1, -6), (2, 2), (9 + 2, 4)])
    # all twelve pieces on the board
    E["all_twelve"] = put([(0, 1), (1, 2), (2, -1), (3, -2), (9 + 0, 3), (9 + 4, 4), (9 + 5, -3), (9 + 6, -4),
                           (18 + 0, 5), (18 + 7, 6), (18 + 8, -5), (18 + 4, -6)])
    # self-gobble: own medium over own small
    E["self_gobble"] = put([(4, 1), (9 + 4, 3)])
    # uncovering hands the opponent a line: P1 large on pos 1 covers P2 medium; P2 has tops at 0 and 2
    E["uncover_loss"] = put([(18 + 1, 5), (9 + 1, -3), (9 + 0, -4), (18 + 2, -5)])
    E["empty"] = np.zeros(27, np.int8)
    return E


# --------------------------------------------------------------------------------------
class _Capture:
    """Stand-in for np.random.choice used only by the greedy harness: records the
    candidate list the reference hands to it and returns its first element."""

    def __init__(self):
        self.calls = []

    def __call__(self, a, *args, **kwargs):
        lst = [int(x) for x in a]
        self.calls.append(lst)
        return np.int64(lst[0])


def greedy_decision(obs, mask, depth):
    """(chosen_before_fallback | -1, actions_depth1 list, returned) with empty history."""
    cap = _Capture()
    orig = np.random.choice
    np.random.choice = cap
    try:
        pol = GreedyGobbletPolicy(depth=depth)
        ret = int(pol.compute_action(obs, mask))
        if cap.calls:  # fallback with chosen None (history is empty)
            return -1, cap.calls[0], ret
        agent = int(obs[..., 12].max())
        pol2 = GreedyGobbletPolicy(depth=depth)
        pol2.prev_actions[agent] = [ret]  # forces the fallback so the list becomes observable
        cap.calls.clear()
        pol2.compute_action(obs, mask)
        assert len(cap.calls) == 1
        return ret, cap.calls[0], ret
    finally:
        np.random.choice = orig


def greedy_vectors(games, n_positions=320, seed=7):
    rng = np.random.default_rng(seed)
    idx = np.flatnonzero((games["done"] == 0) & (games["ply"] >= 1))
    pick = rng.choice(idx, size=min(n_positions, len(idx)), replace=False)
    positions = [(games["squares_after"][i].copy(), int(games["to_move_after"][i])) for i in pick]
    # the immediate-win-overwritten quirk of SURVEY.md App. B: P1 has tops on 0,1 and can finish on 8? build one:
    quirk = np.zeros(27, np.int8)
    quirk[18 + 0] = 5; quirk[9 + 4] = 3   # P1 large at 0, P1 medium at 4 -> diagonal 0,4,8 open at 8
    quirk[0 + 1] = -1; quirk[0 + 3] = -2  # two P2 smalls that threaten nothing
    positions.append((quirk, 0))
    positions.append((-quirk, 1))
    # dense random valid non-terminal boards
    tries = 0
    while len(positions) < n_positions + 2 + 48 and tries < 10000:
        tries += 1
        sq = random_valid_board(rng)
        b = Board(); b.squares = sq.astype(np.float64)
        if b.check_for_winner() == 0:
            positions.append((sq, int(rng.integers(0, 2))))
    env = raw_env()
    env.reset()
    rows = {k: [] for k in ["squares", "to_move", "obs", "mask", "chosen_d1", "cands_d1", "chosen_d2", "cands_d2"]}
    for n, (sq, tm) in enumerate(positions):
        b = Board(); b.squares = sq.astype(np.float64)
        env.board = b
        env.agent_selection = env.agents[tm]
        o = env.observe(env.agents[tm])
        obs, mask = o["observation"], o["action_mask"]
        rows["squares"].append(sq); rows["to_move"].append(tm); rows["obs"].append(obs); rows["mask"].append(mask)
        for depth in (1, 2):
            chosen, cands, _ = greedy_decision(obs, mask, depth)
            cm = np.zeros(54, np.int8); cm[cands] = 1
            assert cands == sorted(cands)
            rows[f"chosen_d{depth}"].append(chosen); rows[f"cands_d{depth}"].append(cm)
        if n % 50 == 0:
            print("  greedy", n, "/", len(positions), flush=True)
    out = {k: np.asarray(v).astype(np.int8) for k, v in rows.items()}
    np.savez_compressed(os.path.join(OUT, "greedy.npz"), **out)
    print(f"greedy.npz: {len(positions)} positions; depth2 fallback(None) {int((out['chosen_d2'] < 0).sum())}; "
          f"quirk chosen d1={out['chosen_d1'][len(pick)]} d2={out['chosen_d2'][len(pick)]}")


# --------------------------------------------------------------------------------------
def render_text():
    """stdout of the reference's render() in "text" / "text_full" mode along two golden games."""
    import contextlib
    import io
    g = np.load(os.path.join(OUT, "random_games.npz"))
    out = []
    for mode in ("text", "text_full"):
        env = raw_env(render_mode=mode)
        for i in np.flatnonzero(g["game"] < 2):
            if g["ply"][i] == 0:
                env.reset()
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                env.step(int(g["action"][i]))  # step() calls render() when a mode is set (gobblet.py:272-273)
            out.append({"mode": mode, "index": int(i), "text": buf.getvalue()})
    with open(os.path.join(OUT, "render_text.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("render_text.json:", len(out), "frames")


def print_pieces_text(n_boards=48):
    """stdout of the reference's debug printers on fixture boards: Board.print_pieces (board.py:223-239),
    Board.print (:155-156) and str(Board) (:241-242); plus render() frames of a raw_env created with
    args.debug = True (gobblet.py:315-316: print_pieces, then the "text" frame) along one golden game."""
    import contextlib
    import io
    bf = np.load(os.path.join(OUT, "board_functions.npz"))
    out = {"boards": [], "debug_frames": []}
    for i in range(n_boards):
        b = Board()
        b.squares = bf["squares"][i].astype(np.float64)
        rec = {"index": i, "squares": bf["squares"][i].astype(int).tolist()}
        for key, fn in (("print_pieces", b.print_pieces), ("print", b.print), ("str", lambda: print(str(b)))):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                fn()
            rec[key] = buf.getvalue()
        out["boards"].append(rec)
    g = np.load(os.path.join(OUT, "random_games.npz"))
    env = raw_env(render_mode="text", args=types.SimpleNamespace(debug=True))
    for i in np.flatnonzero(g["game"] < 1):
        if g["ply"][i] == 0:
            env.reset()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            env.step(int(g["action"][i]))
        out["debug_frames"].append({"index": int(i), "text": buf.getvalue()})
    with open(os.path.join(OUT, "print_pieces.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("print_pieces.json:", len(out["boards"]), "boards,", len(out["debug_frames"]), "debug frames")


def debug_illegal_frames(n_games=12, seed=2024):
    """stdout of raw_env.step with args.debug = True along games that CONTAIN illegal plies (gobblet.py:238-242: the four
    lines "piece: / piece_size: / pos: / --ERROR-- ILLEGAL MOVE").  The legality test of that branch is handed the agent
    NAME as agent_index (``self.board.is_legal(action, self.agent_selection)``), and Board.is_legal compares
    ``agent_index == 0`` (board.py:86), so the test is ALWAYS player_2's: the lines also appear on legal moves of
    player_1 that player_2 could not make, and are missing on illegal moves of player_1 that player_2 could.  Actions:
    seeded; one ply in three is drawn from all 54 actions (often illegal: a silent no-op, the turn passes,
    gobblet.py:244-246), the others from the mover's legal moves.  Also records the screen_width / screen_height
    attributes (gobblet.py:165-166) for a default and an explicit args."""
    import contextlib
    import io
    rng = np.random.default_rng(seed)
    frames = []
    env = raw_env(render_mode="text", args=types.SimpleNamespace(debug=True))
    for game in range(n_games):
        env.reset()
        for ply in range(40):
            agent = env.agent_selection
            if env.terminations[agent]:
                break
            mask = env.observe(agent)["action_mask"]
            legal = np.flatnonzero(mask)
            action = int(rng.integers(0, 54)) if rng.random() < 1 / 3 else int(rng.choice(legal))
            idx = env.agents.index(agent)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                env.step(action)
            frames.append({"game": game, "ply": ply, "agent": idx, "action": action, "legal_for_mover": bool(mask[action]),
                           "text": buf.getvalue(), "squares": env.board.squares.astype(int).tolist()})
    n_err = sum("--ERROR-- ILLEGAL MOVE" in f["text"] for f in frames)
    n_ill = sum(not f["legal_for_mover"] for f in frames)
    quirk_a = sum(f["legal_for_mover"] and "--ERROR--" in f["text"] for f in frames)
    quirk_b = sum((not f["legal_for_mover"]) and "--ERROR--" not in f["text"] for f in frames)
    attrs = {}
    for name, args in (("default", None), ("explicit", types.SimpleNamespace(screen_width=480))):
        e = raw_env(render_mode=None, args=args)
        attrs[name] = {"screen_width": int(e.screen_width), "screen_height": int(e.screen_height)}
    with open(os.path.join(OUT, "debug_illegal.json"), "w") as f:
        json.dump({"frames": frames, "attrs": attrs}, f, indent=0)
    print("debug_illegal.json:", len(frames), "frames;", n_ill, "illegal plies;", n_err, "with the ERROR lines;",
          quirk_a, "legal moves flagged (tested as player_2);", quirk_b, "illegal moves not flagged")


def c1_thousand_games(n_games=1000):
    """the AEC loop of examples/example_basic.py:50-67 over the reference
    raw_env, masked-uniform actions drawn from numpy.random.default_rng(0); whole trajectories."""
    rng = np.random.default_rng(0)
    env = raw_env()
    acts, sq, masks, winners, lens = [], [], [], [], []
    for g in range(n_games):
        env.reset()
        n = 0
        while not env.terminations["player_1"]:
            mask = env.observe(env.agent_selection)["action_mask"]
            a = int(rng.choice(np.arange(len(mask)), p=mask / np.sum(mask)))
            env.step(a)
            acts.append(a); sq.append(i8(env.board.squares)); masks.append(mask.copy())
            winners.append(env.board.check_for_winner())
            n += 1
        lens.append(n)
        if g % 100 == 0:
            print("  c1 game", g, flush=True)
    np.savez_compressed(os.path.join(OUT, "c1_1000_games.npz"), action=np.array(acts, np.int8),
                        squares_after=np.array(sq, np.int8), mask_before=np.packbits(np.array(masks, np.uint8), axis=1),
                        winner=np.array(winners, np.int8), game_len=np.array(lens, np.int16))
    print(f"c1_1000_games.npz: {n_games} games, {len(acts)} plies, mean length {np.mean(lens):.2f}, "
          f"P1 wins {np.mean(np.array(winners)[np.cumsum(lens) - 1] == 1):.3f}")


def greedy_restricted_masks(n_positions=260, seed=21):
    """Greedy decisions when the policy is handed only a SUBSET of the legal moves (1-3 actions, or a
    random 40 %): drives the `len(actions_depth1) > 1` guards and the early `break`s of
    greedy_policy.py:98-101,132-136 that full masks rarely reach."""
    rng = np.random.default_rng(seed)
    g = np.load(os.path.join(OUT, "random_games.npz"))
    idx = np.flatnonzero((g["done"] == 0) & (g["ply"] >= 2))
    pick = rng.choice(idx, size=n_positions, replace=False)
    env = raw_env()
    env.reset()
    rows = {k: [] for k in ["squares", "to_move", "mask", "chosen_d1", "cands_d1", "chosen_d2", "cands_d2"]}
    for n, i in enumerate(pick):
        sq, tm = g["squares_after"][i], int(g["to_move_after"][i])
        b = Board(); b.squares = sq.astype(np.float64)
        env.board = b
        env.agent_selection = env.agents[tm]
        o = env.observe(env.agents[tm])
        legal = np.flatnonzero(o["action_mask"])
        if n % 4 == 3:
            keep = legal[rng.random(len(legal)) < 0.4]
            if len(keep) == 0:
                keep = legal[:1]
        else:
            keep = rng.choice(legal, size=min(len(legal), 1 + n % 3), replace=False)
        mask = np.zeros(54, np.int8); mask[keep] = 1
        rows["squares"].append(sq); rows["to_move"].append(tm); rows["mask"].append(mask)
        for depth in (1, 2):
            chosen, cands, _ = greedy_decision(o["observation"], mask, depth)
            cm = np.zeros(54, np.int8); cm[cands] = 1
            rows[f"chosen_d{depth}"].append(chosen); rows[f"cands_d{depth}"].append(cm)
    out = {k: np.asarray(v).astype(np.int8) for k, v in rows.items()}
    np.savez_compressed(os.path.join(OUT, "greedy_restricted.npz"), **out)
    print(f"greedy_restricted.npz: {n_positions} positions; depth-2 None {int((out['chosen_d2'] < 0).sum())}, "
          f"mask sizes {np.bincount(out['mask'].sum(1))[:5]}")


def greedy_depth3():
    """depth=3 decisions of the reference for every position of greedy.npz and greedy_restricted.npz.
    The depth-3 block (greedy_policy.py:160-208) can only re-assign `chosen_action = action`, which
    :157 has just assigned, so the decisions must equal the depth-2 ones; this fixture pins that on the
    reference itself instead of on a reading of its source."""
    env = raw_env()
    env.reset()
    out = {}
    for tag, name in (("full", "greedy.npz"), ("restricted", "greedy_restricted.npz")):
        g = np.load(os.path.join(OUT, name))
        chosen, cands = [], []
        # the reference's depth 3 costs up to a minute per full-mask position: sample
        index = sorted(set(range(0, len(g["squares"]), 3 if tag == "full" else 2)) | ({320, 321} if tag == "full" else set()))
        for n, i in enumerate(index):
            tm = int(g["to_move"][i])
            b = Board(); b.squares = g["squares"][i].astype(np.float64)
            env.board = b
            env.agent_selection = env.agents[tm]
            obs = env.observe(env.agents[tm])["observation"]
            c, lst, _ = greedy_decision(obs, g["mask"][i], 3)
            cm = np.zeros(54, np.int8); cm[lst] = 1
            chosen.append(c); cands.append(cm)
            if n % 20 == 0:
                print("  depth3", tag, n, "/", len(index), flush=True)
        out[f"index_{tag}"] = np.asarray(index, np.int32)
        out[f"chosen_d3_{tag}"] = np.asarray(chosen, np.int8)
        out[f"cands_d3_{tag}"] = np.asarray(cands, np.int8)
        print(f"{name}: depth-3 == depth-2 on {int((out[f'chosen_d3_{tag}'] == g['chosen_d2'][index]).sum())} of "
              f"{len(chosen)} positions; candidate lists equal: {bool((out[f'cands_d3_{tag}'] == g['cands_d2'][index]).all())}")
    np.savez_compressed(os.path.join(OUT, "greedy_depth3.npz"), **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "greedy_restricted":
        greedy_restricted_masks()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "greedy_depth3":
        greedy_depth3()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "render":
        render_text()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "print_pieces":
        print_pieces_text()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "debug_illegal":
        debug_illegal_frames()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "c1":
        c1_thousand_games()
        return
    kat_collector()
    games = random_games()
    rng = np.random.default_rng(99)
    E = edge_cases()
    names = list(E)
    boards = [E[k] for k in names] + [random_valid_board(rng) for _ in range(400)]
    bf = board_fn_vectors(boards)
    bf["edge_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "board_functions.npz"), **bf)
    print("board_functions.npz:", len(boards), "boards; winners", dict(zip(*np.unique(bf["winner"], return_counts=True))))
    for k in names:
        i = names.index(k)
        print(f"  {k}: winner {bf['winner'][i]} flat {bf['flatboard'][i].tolist()}")
    greedy_vectors(games)
    render_text()
    print_pieces_text()
    debug_illegal_frames()
    c1_thousand_games()
    greedy_restricted_masks()
    greedy_depth3()


if __name__ == "__main__":
    main()
