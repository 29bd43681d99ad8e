"""ctypes wrapper of oracle/liboracle.so — the CPU oracle.  TEST INFRASTRUCTURE ONLY (see orc.h):
imported by tests/, __graft_entry__.smoke(), bench.py (cpu_baseline leg + the parity spot checks outside its timed
regions) and tools/whole_games_config3.py's end-of-run check, never by the product package."""
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
        tmp = f"liboracle.{os.getpid()}.tmp.so"   # private output + atomic rename: ranks may build concurrently
        r = subprocess.run(["make", "-C", HERE, "-B", "liboracle.so", f"OUT={tmp}"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
        os.replace(os.path.join(HERE, tmp), LIB_PATH)
    return LIB_PATH


class OrcEnv(Structure):
    _fields_ = [("black", c_uint64), ("white", c_uint64), ("next_player", c_int), ("turn", c_int),
                ("done", c_int), ("winner", c_int), ("ended_illegal", c_int), ("ended_resign", c_int)]


_lib = None
_lock = __import__("threading").RLock()   # first use may come from several threads at once (tests, bench): build + bind exactly once


def load():
    global _lib
    with _lock:
        return _load_locked()


def _load_locked():
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


# ---- raz-rng-v1 / raz-math-v1 / raznet-forward-v1 / MCTS self-play (orc_rng.c, orc_net.c, orc_mcts.c) ----
from ctypes import c_double, c_float, c_uint32, c_longlong, c_size_t, c_char_p  # noqa: E402


class OrcPlayCfg(Structure):
    _fields_ = [("thinking_loop", c_int), ("required_visit_to_decide_action", c_int),
                ("start_rethinking_turn", c_int), ("c_puct", c_double), ("noise_eps", c_double),
                ("dirichlet_alpha", c_double), ("change_tau_turn", c_int), ("virtual_loss", c_int),
                ("parallel_search_num", c_int), ("has_resign_threshold", c_int),
                ("resign_threshold", c_double), ("allowed_resign_turn", c_int),
                ("disable_resignation_rate", c_double), ("use_solver_turn", c_int),
                ("use_solver_turn_in_simulation", c_int), ("share_mtcs_info", c_int),
                ("save_policy_of_tau_1", c_int)]


class OrcPlyRecord(Structure):
    _fields_ = [("player", c_int), ("turn", c_int), ("own", c_uint64), ("enemy", c_uint64),
                ("action", c_int), ("solved", c_int), ("has_row", c_int), ("sims", c_int), ("loops", c_int),
                ("n", c_double), ("q", c_double), ("root_n", c_double * 64), ("root_w", c_double * 64),
                ("saved_policy", c_double * 64)]


class OrcGameSummary(Structure):
    _fields_ = [("winner", c_int), ("turn", c_int), ("plies", c_int), ("enable_resign", c_int),
                ("resigned_black", c_int), ("resigned_white", c_int), ("black", c_uint64),
                ("white", c_uint64), ("drop_draw_u", c_double), ("n_sims", c_longlong),
                ("n_expand", c_longlong), ("n_mirror_hits", c_longlong), ("n_terminal", c_longlong),
                ("n_nodes", c_longlong), ("n_solved_leaves", c_longlong), ("n_solver_nodes", c_longlong),
                ("n_parked", c_longlong)]


def play_cfg_from_config(config, parallel_search_num=1):
    """OrcPlayCfg from a Config-like object (reference Config or reversi_alpha_zero_amd Config)."""
    p, pd = config.play, config.play_data
    return OrcPlayCfg(
        thinking_loop=p.thinking_loop, required_visit_to_decide_action=p.required_visit_to_decide_action,
        start_rethinking_turn=p.start_rethinking_turn, c_puct=float(p.c_puct), noise_eps=float(p.noise_eps),
        dirichlet_alpha=float(p.dirichlet_alpha), change_tau_turn=p.change_tau_turn,
        virtual_loss=p.virtual_loss, parallel_search_num=parallel_search_num,
        has_resign_threshold=int(p.resign_threshold is not None),
        resign_threshold=float(p.resign_threshold if p.resign_threshold is not None else 0.0),
        allowed_resign_turn=p.allowed_resign_turn, disable_resignation_rate=float(p.disable_resignation_rate),
        use_solver_turn=int(p.use_solver_turn or 0), use_solver_turn_in_simulation=int(p.use_solver_turn_in_simulation or 0),
        share_mtcs_info=int(bool(p.share_mtcs_info_in_self_play)), save_policy_of_tau_1=int(bool(pd.save_policy_of_tau_1)))


# the reference's NN seam as a C callback (orc.h orc_nn_fn)
NN_FN = ctypes.CFUNCTYPE(None, c_void_p, c_uint64, c_uint64, POINTER(c_float), POINTER(c_float))

_ext_done = False


def load_ext():
    """load() + signatures for the rng / net / mcts entry points."""
    with _lock:
        return _load_ext_locked()


def _load_ext_locked():
    global _ext_done
    lib = load()
    if not _ext_done:
        lib.orc_rng_pair.argtypes = [c_uint32] * 6 + [POINTER(c_double * 2)]
        lib.orc_rng_pair.restype = None
        for n, a in (("orc_det_log", [c_double]), ("orc_det_exp", [c_double]), ("orc_det_pow", [c_double, c_double])):
            getattr(lib, n).argtypes = a
            getattr(lib, n).restype = c_double
        for n in ("orc_det_expf", "orc_det_tanhf"):
            getattr(lib, n).argtypes = [c_float]
            getattr(lib, n).restype = c_float
        lib.orc_det_cos2.argtypes = [c_double]
        lib.orc_det_cos2.restype = c_double
        lib.orc_dirichlet_gammas.argtypes = [c_double, c_int, c_uint32, c_uint32, c_uint32, c_void_p]
        lib.orc_dirichlet_gammas.restype = None
        lib.orc_gamma_sample.argtypes = [c_double] + [c_uint32] * 4
        lib.orc_gamma_sample.restype = c_double
        lib.orc_dirichlet_noise_of_mask.argtypes = [c_uint64, c_double, c_uint32, c_uint32, c_uint32, POINTER(c_double * 64)]
        lib.orc_dirichlet_noise_of_mask.restype = None
        lib.orc_net_forward_planes.argtypes = [c_char_p, c_size_t, c_void_p, c_void_p, c_void_p]
        lib.orc_net_forward.argtypes = [c_char_p, c_size_t, c_uint64, c_uint64, c_void_p, c_void_p]
        lib.orc_solver_new.restype = c_void_p
        lib.orc_solver_free.argtypes = [c_void_p]
        lib.orc_solver_nodes.argtypes = [c_void_p]
        lib.orc_solver_nodes.restype = c_longlong
        lib.orc_solver_solve.argtypes = [c_void_p, c_uint64, c_uint64, c_int, c_int, POINTER(c_int), POINTER(c_int)]
        lib.orc_selfplay_game.argtypes = [POINTER(OrcPlayCfg), c_char_p, c_size_t, c_uint32, c_uint32, c_int,
                                          POINTER(OrcPlyRecord), c_int, POINTER(OrcGameSummary)]
        lib.orc_tree_new.restype = c_void_p
        lib.orc_tree_free.argtypes = [c_void_p]
        lib.orc_selfplay_game_on.argtypes = [c_void_p] + lib.orc_selfplay_game.argtypes
        lib.orc_selfplay_game_ex.argtypes = [c_void_p, POINTER(OrcPlayCfg), c_char_p, c_size_t, NN_FN, c_void_p, c_uint32,
                                             c_uint32, c_int, POINTER(OrcPlyRecord), c_int, c_int, POINTER(OrcGameSummary)]
        lib.orc_selfplay_game_from.argtypes = [POINTER(OrcPlayCfg), c_char_p, c_size_t, NN_FN, c_void_p, c_uint32, c_uint32, c_int,
                                               c_uint64, c_uint64, c_int, POINTER(OrcPlyRecord), c_int, c_int, POINTER(OrcGameSummary)]
        _ext_done = True
    return lib


def rng_pair(seed, game, purpose, event, sub=0, idx=0):
    lib = load_ext()
    out = (c_double * 2)()
    lib.orc_rng_pair(seed, game, purpose, event, sub, idx, ctypes.byref(out))
    return out[0], out[1]


def net_forward_planes(blob, planes):
    """planes: (N,2,8,8) or (2,8,8) array-like of 0/1 -> (policy (N,64) f32, value (N,1) f32)."""
    import numpy as np
    lib = load_ext()
    x = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 128)
    pol = np.zeros((x.shape[0], 64), dtype=np.float32)
    val = np.zeros((x.shape[0], 1), dtype=np.float32)
    for i in range(x.shape[0]):
        rc = lib.orc_net_forward_planes(blob, len(blob), x[i].ctypes.data, pol[i].ctypes.data, val[i].ctypes.data)
        if rc != 0:
            raise ValueError("bad raznet blob")
    return pol, val


class Tree:
    """The MCTSInfo one worker carries across `reset_mtcs_info_per_game` games (worker/self_play.py:109-111,
    132-134): pass the same Tree to consecutive selfplay_game calls."""

    def __init__(self):
        self._lib = load_ext()
        self.h = self._lib.orc_tree_new()

    def __del__(self):
        if getattr(self, "h", None):
            self._lib.orc_tree_free(self.h)
            self.h = None


def selfplay_game(cfg, blob, seed, game_id, sims_per_move, max_plies=128, tree=None, nn=None, stop_after_plies=0, start=None):
    """Run one oracle self-play game.  Returns (list of ply dicts, summary dict).
    nn (optional): callable (own, enemy) -> (policy64 float32 array, value float): the injected NN seam
    (ReversiPlayer(api=...)); blob may then be None.  stop_after_plies > 0: only the game's first plies.
    start (optional): (black, white, next_player) - take the game up at that position (orc_selfplay_game_from)."""
    import numpy as np
    lib = load_ext()
    plies = (OrcPlyRecord * max_plies)()
    summ = OrcGameSummary()
    if nn is not None:
        def _cb(ctx, own, enemy, pol, val):
            p, v = nn(int(own), int(enemy))
            p = np.ascontiguousarray(p, dtype=np.float32)
            ctypes.memmove(pol, p.ctypes.data, 256)
            val[0] = float(v)
        cb = NN_FN(_cb)
    else:
        cb = ctypes.cast(None, NN_FN)
    if start is not None:
        assert tree is None
        n = lib.orc_selfplay_game_from(ctypes.byref(cfg), blob, len(blob) if blob else 0, cb, None, seed, game_id, sims_per_move,
                                       int(start[0]), int(start[1]), int(start[2]), plies, max_plies, stop_after_plies, ctypes.byref(summ))
    else:
        n = lib.orc_selfplay_game_ex(tree.h if tree is not None else None, ctypes.byref(cfg), blob, len(blob) if blob else 0,
                                     cb, None, seed, game_id, sims_per_move, plies, max_plies, stop_after_plies,
                                     ctypes.byref(summ))
    if n < 0:
        raise RuntimeError("orc_selfplay_game failed (unsupported config or too many plies)")
    out = []
    for i in range(n):
        r = plies[i]
        out.append({"player": r.player, "turn": r.turn, "own": r.own, "enemy": r.enemy, "action": r.action,
                    "solved": bool(r.solved), "has_row": bool(r.has_row), "sims": r.sims, "loops": r.loops, "n": r.n, "q": r.q,
                    "root_n": list(r.root_n), "root_w": list(r.root_w), "saved_policy": list(r.saved_policy)})
    s = {k: getattr(summ, k) for k, _ in OrcGameSummary._fields_}
    return out, s
