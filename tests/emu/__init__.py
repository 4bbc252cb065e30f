"""Host emulation of the device header (TEST INFRASTRUCTURE, see emu_device.cpp)."""
import ctypes as C
import fcntl
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgobblet_emu.so")
_SRCS = [os.path.join(_HERE, "emu_device.cpp"), os.path.join(_HERE, "greedy_root_rule.h"),
         os.path.join(_HERE, "..", "..", "gobblet-rl_amd", "csrc", "gobblet_device.h")]


def build():
    with open(_LIB + ".lock", "w") as lock:  # pytest-xdist workers may arrive together
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(_LIB) or any(os.path.getmtime(_LIB) < os.path.getmtime(s) for s in _SRCS):
                subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-fPIC", "-shared",
                                       "-o", _LIB, _SRCS[0]],
                                      env={k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB")
                                           and not k.startswith(("ROCP", "ROCPROFILER", "ROCTRACER", "ROCTX"))})
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def legal_mask(state, to_move):
    n = len(state); out = np.full((n, 54), 77, np.int8)
    lib().emu_legal_mask(_p(state), _p(to_move), _p(out), C.c_int64(n)); return out


def is_legal(state, agent, actions):
    n = len(state); out = np.full(n, 77, np.int8); actions = np.ascontiguousarray(actions, np.int32)
    lib().emu_is_legal(_p(state), _p(agent), _p(actions), _p(out), C.c_int64(n)); return out


def play_turn(state, agent, actions):
    actions = np.ascontiguousarray(actions, np.int32)
    lib().emu_play_turn(_p(state), _p(agent), _p(actions), C.c_int64(len(state)))


def planes_roundtrip(state):
    """state rows -> bit planes -> rows (planes_to_row: how k_collect5 writes a group's boards back at the end of a launch)."""
    n = len(state); out = np.full((n, 27), 77, np.int8)
    lib().emu_planes_roundtrip(_p(state), _p(out), C.c_int64(n)); return out


def winner(state):
    n = len(state); out = np.full(n, 77, np.int8)
    lib().emu_winner(_p(state), _p(out), C.c_int64(n)); return out


def flatboard(state):
    n = len(state); out = np.full((n, 9), 77, np.int8)
    lib().emu_flatboard(_p(state), _p(out), C.c_int64(n)); return out


def covered(state):
    n = len(state); out = np.full((n, 27), 77, np.int8)
    lib().emu_covered(_p(state), _p(out), C.c_int64(n)); return out


def observe(state, to_move, agent_sel=-1):
    n = len(state); out = np.full((n, 3, 3, 13), 77, np.int8)
    lib().emu_observe(_p(state), _p(to_move), C.c_int(agent_sel), _p(out), C.c_int64(n)); return out


def step(state, to_move, done, actions, illegal_mode=0, auto_reset=False):
    n = len(state); actions = np.ascontiguousarray(actions, np.int32)
    w = np.full(n, 77, np.int8); rw = np.full((n, 2), 77, np.int8)
    mask = np.full((n, 54), 77, np.int8); obs = np.full((n, 3, 3, 13), 77, np.int8)
    lib().emu_step(_p(state), _p(to_move), _p(done), _p(actions), _p(w), _p(rw), _p(mask), _p(obs), C.c_int64(n),
                   C.c_int(illegal_mode), C.c_int(int(auto_reset)))
    return {"winner": w, "reward": rw, "mask": mask, "obs": obs}


def sample(mask, seed, env_base, ply):
    n = len(mask); out = np.full(n, 77, np.int32)
    lib().emu_sample(_p(mask), _p(out), C.c_int64(n), C.c_uint64(seed), C.c_uint64(env_base), C.c_uint32(ply))
    return out


def decode_obs(obs):
    n = len(obs); st = np.full((n, 27), 77, np.int8); tm = np.full(n, 77, np.int8)
    lib().emu_decode_obs(_p(obs), _p(st), _p(tm), C.c_int64(n)); return st, tm


def rollout(state, to_move, done, seed, env_base, ply0, plies, illegal_mode=0):
    n = len(state)
    a = np.full(n, 77, np.int32); w = np.full(n, 77, np.int8); rw = np.full((n, 2), 77, np.int8)
    mask = np.full((n, 54), 77, np.int8); obs = np.full((n, 3, 3, 13), 77, np.int8); cnt = np.zeros(4, np.int64)
    lib().emu_rollout(_p(state), _p(to_move), _p(done), _p(a), _p(w), _p(rw), _p(mask), _p(obs), C.c_int64(n),
                      C.c_uint64(seed), C.c_uint64(env_base), C.c_uint32(ply0), C.c_uint32(plies),
                      C.c_int(illegal_mode), _p(cnt))
    return {"actions": a, "winner": w, "reward": rw, "mask": mask, "obs": obs, "counters": cnt}


def rollout_small(state, to_move, done, seed, env_base, ply0, plies, illegal_mode=0, lpb=4):
    """The role kernel's walk (sub-tiles of 64 / lpb boards, lpb lanes per board): outputs of the last ply."""
    n = len(state)
    a = np.full(n, 77, np.int32); w = np.full(n, 77, np.int8); rw = np.full((n, 2), 77, np.int8)
    mask = np.full((n, 54), 77, np.int8); obs = np.full((n, 3, 3, 13), 77, np.int8)
    lib().emu_rollout_small(_p(state), _p(to_move), _p(done), _p(a), _p(w), _p(rw), _p(mask), _p(obs), C.c_int64(n),
                            C.c_uint64(seed), C.c_uint64(env_base), C.c_uint32(ply0), C.c_uint32(plies), C.c_int(illegal_mode),
                            C.c_int(lpb))
    return {"actions": a, "winner": w, "reward": rw, "mask": mask, "obs": obs}


def greedy(state, to_move, mask=None, hist=None, depth=2, pooled=True):
    """pooled=True: the kernel's flow (pairs pooled over a tile + greedy_replay_sets); False: greedy_decide."""
    n = len(state)
    act = np.full(n, 77, np.int32); cm = np.full((n, 54), 77, np.int8); fb = np.full(n, 77, np.int8)
    lib().emu_greedy(_p(state), _p(to_move), _p(mask), _p(hist), C.c_int(depth), _p(act), _p(cm), _p(fb), C.c_int64(n),
                     C.c_int(int(pooled)), None, None, C.c_uint64(0), C.c_uint64(0), C.c_uint32(0))
    return act, cm, fb


def validate(state):
    n = len(state); out = np.full(n, 77, np.int8)
    lib().emu_validate(_p(state), _p(out), C.c_int64(n)); return out


def greedy_act(state, to_move, hist, seed, env_base, call, mask=None, depth=2):
    """gbl_greedy_act: (returned action, chosen-or--1, candidate mask, fallback); hist (n,2,3) updated in place."""
    n = len(state)
    out = np.full(n, 77, np.int32); act = np.full(n, 77, np.int32); cm = np.full((n, 54), 77, np.int8)
    fb = np.full(n, 77, np.int8)
    lib().emu_greedy(_p(state), _p(to_move), _p(mask), None, C.c_int(depth), _p(act), _p(cm), _p(fb), C.c_int64(n),
                     C.c_int(1), _p(hist), _p(out), C.c_uint64(seed), C.c_uint64(env_base), C.c_uint32(call))
    return out, act, cm, fb


def greedy_root_rule(state, to_move, mask=None):
    """tests/emu/greedy_root_rule.h against the exact evaluation: (boards, candidates settled from the root, of them with
    a winning reply, placements left to the exact evaluation, mismatches)."""
    n = len(to_move)
    o = np.zeros(5, np.int64)
    lib().emu_greedy_root_rule(_p(np.ascontiguousarray(state)), _p(np.ascontiguousarray(to_move)),
                               None if mask is None else _p(np.ascontiguousarray(mask)), C.c_int64(n), _p(o))
    return tuple(int(x) for x in o)


def greedy_stats():
    """(pairs evaluated, of them pairs for which reply_is_plain does not hold -- the ordered line steps --, cross-check
    failures -- fast form != ordered form of a pair, closed form != loop form of the depth-2 loop, a settled placement whose
    table summary differs from its evaluation --, placements settled from the root's replies, items dealt out) since the
    library was loaded."""
    o = np.zeros(5, np.int64)
    lib().emu_greedy_stats(_p(o)); return tuple(int(x) for x in o)


def greedy_vroot_rule(state, to_move, mask=None, cap=6):
    """The virtual-root rule for moves of placed pieces (tests/emu/greedy_root_rule.h) against the exact evaluation:
    (boards, pairs evaluated today, of them moves of placed pieces, virtual roots used, candidates settled, candidates left,
    items dealt out, MISMATCHES)."""
    o = np.zeros(8, np.int64)
    lib().emu_greedy_vroot_rule(_p(np.ascontiguousarray(state)), _p(np.ascontiguousarray(to_move)),
                                _p(np.ascontiguousarray(mask)) if mask is not None else None, len(state), int(cap), _p(o))
    return tuple(int(x) for x in o)
