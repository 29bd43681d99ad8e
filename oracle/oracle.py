"""ctypes wrapper of oracle/liboracle.so — the CPU oracle.  TEST INFRASTRUCTURE ONLY (see orc.h):
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
import ctypes
import os
import subprocess
from ctypes import c_int, c_uint64, c_void_p, POINTER, Structure

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")


def build(force=False):
    """gcc-compile the oracle (seconds)."""
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.startswith("orc") and f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(LIB_PATH) or \
        any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-C", HERE, "-B", "liboracle.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


class OrcEnv(Structure):
    _fields_ = [("black", c_uint64), ("white", c_uint64), ("next_player", c_int), ("turn", c_int),
                ("done", c_int), ("winner", c_int), ("ended_illegal", c_int), ("ended_resign", c_int)]


_lib = None


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB_PATH)
        lib.orc_find_correct_moves.restype = c_uint64
        lib.orc_find_correct_moves.argtypes = [c_uint64, c_uint64]
        lib.orc_calc_flip.restype = c_uint64
        lib.orc_calc_flip.argtypes = [c_int, c_uint64, c_uint64]
        lib.orc_bit_count.restype = c_int
        lib.orc_bit_count.argtypes = [c_uint64]
        for n in ("orc_flip_vertical", "orc_flip_diag_a1h8", "orc_rotate90", "orc_rotate180"):
            getattr(lib, n).restype = c_uint64
            getattr(lib, n).argtypes = [c_uint64]
        lib.orc_bit_to_array.restype = None
        lib.orc_bit_to_array.argtypes = [c_uint64, c_int, c_void_p]
        lib.orc_env_reset.argtypes = [POINTER(OrcEnv)]
        lib.orc_env_update.argtypes = [POINTER(OrcEnv), c_uint64, c_uint64, c_int]
        lib.orc_env_step.argtypes = [POINTER(OrcEnv), c_int]
        lib.orc_find_correct_moves_n.argtypes = [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]
        lib.orc_calc_flip_n.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t]
        lib.orc_step_n.argtypes = [c_void_p] * 6 + [ctypes.c_size_t]
        for n in ("orc_find_correct_moves_n", "orc_calc_flip_n", "orc_step_n"):
            getattr(lib, n).restype = None
        _lib = lib
    return _lib


# ---- numpy conveniences over the array helpers ----
def np_find_correct_moves(own, enemy):
    import numpy as np
    lib = load()
    own = np.ascontiguousarray(own, dtype=np.uint64)
    enemy = np.ascontiguousarray(enemy, dtype=np.uint64)
    out = np.empty_like(own)
    lib.orc_find_correct_moves_n(own.ctypes.data, enemy.ctypes.data, out.ctypes.data, own.size)
    return out


def np_calc_flip(pos, own, enemy):
    import numpy as np
    lib = load()
    pos = np.ascontiguousarray(pos, dtype=np.uint8)
    own = np.ascontiguousarray(own, dtype=np.uint64)
    enemy = np.ascontiguousarray(enemy, dtype=np.uint64)
    out = np.empty_like(own)
    lib.orc_calc_flip_n(pos.ctypes.data, own.ctypes.data, enemy.ctypes.data, out.ctypes.data, own.size)
    return out


def np_step(black, white, player, status, action):
    """Returns new (black, white, player, status, legal) arrays."""
    import numpy as np
    lib = load()
    black = np.array(black, dtype=np.uint64)
    white = np.array(white, dtype=np.uint64)
    player = np.array(player, dtype=np.uint8)
    status = np.array(status, dtype=np.uint8)
    action = np.ascontiguousarray(action, dtype=np.uint8)
    legal = np.zeros_like(black)
    lib.orc_step_n(black.ctypes.data, white.ctypes.data, player.ctypes.data, status.ctypes.data,
                   legal.ctypes.data, action.ctypes.data, black.size)
    return black, white, player, status, legal
