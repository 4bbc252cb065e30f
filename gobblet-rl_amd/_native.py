"""ctypes binding of the HIP library (``csrc/libgobblet_hip.so``, C-ABI in ``include/gobblet_hip.h``).

There is no CPU fallback: if the library cannot be built or loaded this module raises.
``import torch`` happens first on purpose -- the PyTorch-ROCm wheel bundles the HIP runtime
under the same SONAME (``libamdhip64.so.7``) the library links against, so importing torch
first makes the process use ONE runtime and lets torch streams / tensors be handed across
the ABI as plain pointers.
"""
from __future__ import annotations

import ctypes as C
import fcntl
import os
import platform
import shutil
import subprocess

import torch  # noqa: F401  (must precede loading the HIP library, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libgobblet_hip.so")
_FOREIGN = False  # use_library() was called: LIB_PATH is somebody's own build, never rebuilt from here
SOURCES = [os.path.join(CSRC, "gobblet_hip.hip"), os.path.join(CSRC, "gobblet_device.h"), os.path.join(CSRC, "gobblet_diag.h"),
           os.path.join(CSRC, "gobblet_knobs.h"), os.path.join(_HERE, "..", "include", "gobblet_hip.h")]
# -amdgpu-kernarg-preload-count: the first 16 dwords of a kernel's arguments arrive in SGPRs with the wave
# launch (the kernels order their arguments for that), so a wavefront's first loads do not wait for a
# kernel-argument fetch; firmware without the feature runs the compiler's fallback prologue
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mcode-object-version=5",
               "-mllvm", "-amdgpu-kernarg-preload-count=16"]

OK, ERR_ARG, ERR_ALIGN, ERR_HIP = 0, -1, -2, -3
ILLEGAL_NOOP, ILLEGAL_TERMINATE = 0, 1
POLICY_RANDOM, POLICY_GREEDY1, POLICY_GREEDY2, POLICY_GREEDY3 = 0, 1, 2, 3  # gbl_collect_policy
HOW_RANDOM, HOW_GREEDY, HOW_FALLBACK = 0, 1, 2
STATUS_ILLEGAL, STATUS_OUT_OF_RANGE = 1, 2  # gbl_step_ex / gbl_collect_from_ex status bits
CELLS, ACTIONS, OBS_BYTES = 27, 54, 117
COUNTER_STRIPES, COUNTER_STRIDE = 64, 16
# gbl_board_eval record (include/gobblet_hip.h GBL_REC_*): field -> (byte offset, bytes)
REC_BYTES = 432
REC_FIELDS = {"squares": (0, 27), "winner": (28, 1), "flat": (32, 9), "covered": (44, 27), "mask0": (72, 54),
              "mask1": (128, 54), "obs0": (184, 117), "obs1": (304, 117)}

# every symbol include/gobblet_hip.h declares: (name, restype, argtypes)
_vp, _i64, _u64, _u32, _int = C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_int
SIGNATURES = {
    "gbl_layout_info": (_int, [C.POINTER(C.c_int32)]),
    "gbl_last_error": (C.c_char_p, []),
    "gbl_reset": (_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "gbl_legal_mask": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "gbl_is_legal": (_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "gbl_play_turn": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "gbl_winner": (_int, [_vp, _vp, _i64, _vp]),
    "gbl_flatboard": (_int, [_vp, _vp, _i64, _vp]),
    "gbl_covered": (_int, [_vp, _vp, _i64, _vp]),
    "gbl_validate": (_int, [_vp, _vp, _i64, _vp]),
    "gbl_observe": (_int, [_vp, _vp, _int, _vp, _i64, _vp]),
    "gbl_board_eval": (_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "gbl_pinned_alloc": (_int, [_i64, C.POINTER(_vp), C.POINTER(_vp)]),
    "gbl_pinned_free": (_int, [_vp]),
    "gbl_step": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp]),
    "gbl_step_into": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp]),
    "gbl_step_ex": (_int, [_vp] * 14 + [_u64, _u64, _u32, _vp, _i64, _int, _int, _vp]),
    "gbl_sample": (_int, [_vp, _vp, _i64, _u64, _u64, _u32, _vp]),
    "gbl_rollout": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _u64, _u64, _u32, _u32, _int, _vp, _vp, _vp]),
    "gbl_decode_obs": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "gbl_greedy": (_int, [_vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp]),
    "gbl_greedy_act": (_int, [_vp, _vp, _vp, _vp, _int, _u64, _u64, _u32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gbl_sample_at": (_int, [_vp, _vp, _i64, _u64, _u64, _u32, _vp, _vp]),
    "gbl_rollout_at": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _u64, _u64, _u32, _vp, _u32, _int, _vp, _vp, _vp]),
    "gbl_greedy_act_at": (_int, [_vp, _vp, _vp, _vp, _int, _u64, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gbl_counter_add": (_int, [_vp, _u32, _vp]),
    "gbl_collect": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _u64, _u64, _u32, _vp, _u32,
                           _int, _vp, _vp, _vp]),
    "gbl_collect_from": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _u64, _u64, _u32, _vp, _u32,
                                _int, _vp, _vp, _vp]),
    "gbl_collect_from_ex": (_int, [_vp] * 12 + [_i64, _i64, _i64, _u64, _u64, _u32, _vp, _u32, _int, _vp, _vp, _vp]),
    "gbl_collect_policy": (_int, [_vp] * 14 + [_i64, _i64, _i64, _u64, _u64, _u32, _vp, _u32, _int, _int, _int, _int, _vp, _vp, _vp]),
    "gbl_collect_variant": (_int, [_i64, _u32, _int, _int]),
    "gbl_block_alloc": (_int, [_i64, C.POINTER(_vp)]),
    "gbl_block_free": (_int, [_vp]),
    "gbl_device_memory": (_int, [C.POINTER(_i64), C.POINTER(_i64)]),
    "gbl_placement_probe": (_int, [_vp, _i64, _vp, _i64, _i64, _int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), _vp]),
}


class GobbletHipError(RuntimeError):
    pass


def use_library(path: str) -> None:
    """Load a differently built library of the same ABI instead of csrc/libgobblet_hip.so (kernel A/B experiments, diagnostic
    builds: scripts/, tests/conftest.py).  Call it before anything else of the package touches the library.  The product reads
    no environment variable: the experiment scripts do, and call this."""
    global LIB_PATH, _FOREIGN, _lib
    if _lib is not None:
        raise GobbletHipError("use_library() after the library has been loaded")
    LIB_PATH, _FOREIGN = os.path.abspath(path), True


def needs_build() -> bool:
    if _FOREIGN:
        return not os.path.exists(LIB_PATH)
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES)


def _compiler_env() -> dict:
    """The environment for a compiler child process WITHOUT a profiler's preload: under `rocprofv3 --pmc -- python3
    ...` this process is GPU-initialised by the preloaded tool library, and a child that inherits the preload and
    then execs clang is the exec-after-GPU-init pattern that takes a pool machine down."""
    drop = ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE")
    return {k: v for k, v in os.environ.items()
            if k not in drop and not k.startswith(("ROCP", "ROCPROFILER", "ROCTRACER", "ROCTX"))}


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise GobbletHipError("hipcc not found: cannot build csrc/libgobblet_hip.so (there is no CPU fallback)")
    # one builder at a time (ranks of a multi-process launch may all arrive here); the library appears
    # under its final name only when complete
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
                cmd = [hipcc, *HIPCC_FLAGS, "-o", tmp, SOURCES[0]]
                if verbose:
                    print(" ".join(cmd[:-3] + ["-o", LIB_PATH, SOURCES[0]]))
                try:
                    subprocess.check_call(cmd, cwd=CSRC, env=_compiler_env())
                    os.replace(tmp, LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded library with typed entry points. Raises if it is missing and cannot be built."""
    global _lib
    if _lib is None:
        path = build()
        try:
            L = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise GobbletHipError(f"cannot load {path}: {e} (there is no CPU fallback)") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise GobbletHipError(f"{path} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        msg = lib().gbl_last_error().decode("utf-8", "replace")
        raise GobbletHipError(f"{what or 'gobblet_hip'} failed (code {rc}): {msg}")


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device):
    """The caller's HIP stream on `device` as a plain pointer (None on the host flavour: its calls are synchronous)."""
    device = torch.device(device)
    if device.type != "cuda":
        return None
    return torch.cuda.current_stream(device).cuda_stream


# ---- the HOST flavour of the ABI (include/gobblet_cpu.h, csrc/gobblet_cpu.cpp): gbl_cpu_* -------------------------------------
# A flavour the caller ASKS for (device="cpu": BASELINE config 1 "on CPU", bench.py's CPU twin), never a fallback of the HIP
# path: a "cuda" device without the HIP library still raises.  Built with g++ from the same device header.
CPU_LIB_PATH = os.path.join(CSRC, "libgobblet_cpu.so")
CPU_SOURCES = [os.path.join(CSRC, "gobblet_cpu.cpp"), os.path.join(CSRC, "gobblet_device.h"),
               os.path.join(_HERE, "..", "include", "gobblet_cpu.h"), os.path.join(_HERE, "..", "include", "gobblet_hip.h")]
CPU_CXX_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-Wno-attributes"] + \
                (["-mpopcnt"] if platform.machine() in ("x86_64", "AMD64") else [])
_NO_HOST_FLAVOUR = ("gbl_pinned_alloc", "gbl_pinned_free", "gbl_block_alloc", "gbl_block_free", "gbl_device_memory",
                    "gbl_placement_probe", "gbl_collect_variant")
CPU_SIGNATURES = {"gbl_cpu_" + k[4:]: v for k, v in SIGNATURES.items() if k not in _NO_HOST_FLAVOUR}
CPU_SIGNATURES["gbl_cpu_set_threads"] = (_int, [_int])


def build_cpu(force: bool = False) -> str:
    """Compile the host flavour in-tree (g++; no GPU toolchain needed)."""
    def stale():
        return not os.path.exists(CPU_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(CPU_LIB_PATH) for s in CPU_SOURCES)
    if not force and not stale():
        return CPU_LIB_PATH
    # (a box with the GPU toolchain but no g++: ROCm's clang++ compiles the host flavour just as well)
    cxx = shutil.which("g++") or shutil.which("c++") or shutil.which("clang++") or shutil.which("amdclang++") or \
        next((p for p in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/bin/amdclang++") if os.path.exists(p)), None)
    if not cxx:
        raise GobbletHipError("no C++ compiler found: cannot build csrc/libgobblet_cpu.so")
    with open(CPU_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                tmp = f"{CPU_LIB_PATH}.{os.getpid()}.tmp"
                try:
                    subprocess.check_call([cxx, *CPU_CXX_FLAGS, "-o", tmp, CPU_SOURCES[0]], cwd=CSRC, env=_compiler_env())
                    os.replace(tmp, CPU_LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return CPU_LIB_PATH


_cpu_raw = None


def cpu_raw() -> C.CDLL:
    """libgobblet_cpu.so with typed gbl_cpu_* entry points (plain return codes)."""
    global _cpu_raw
    if _cpu_raw is None:
        L = C.CDLL(build_cpu())
        for name, (res, args) in CPU_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _cpu_raw = L
    return _cpu_raw


class _HostFlavour:
    """The host library under the DEVICE entry points' names (``gbl_step`` -> ``gbl_cpu_step``), so that the Python layer above
    is the same code for either flavour.  A failing call raises here, with the host library's own message."""

    def __init__(self, raw):
        self._raw = raw

    def __getattr__(self, name):
        if not name.startswith("gbl_"):
            raise AttributeError(name)
        if name in _NO_HOST_FLAVOUR:
            raise GobbletHipError(f"{name} has no host flavour (device-memory helper)")
        fn = getattr(self._raw, "gbl_cpu_" + name[4:])
        if fn.restype is not _int:
            return fn

        def call(*args):
            rc = fn(*args)
            if rc != OK:
                raise GobbletHipError(f"{name} (host flavour) failed (code {rc}): "
                                      f"{self._raw.gbl_cpu_last_error().decode('utf-8', 'replace')}")
            return rc
        setattr(self, name, call)
        return call


_cpu = None


def lib_for(device):
    """The library that serves `device`: the HIP library for a GPU, the host flavour for "cpu" (asked for, never fallen back to)."""
    global _cpu
    if torch.device(device).type == "cuda":
        return lib()
    if torch.device(device).type != "cpu":
        raise GobbletHipError(f"no flavour of the library serves device {device!r}")
    if _cpu is None:
        _cpu = _HostFlavour(cpu_raw())
    return _cpu
