"""ctypes binding of csrc/libraz.so (C ABI declared in include/raz.h).

This is the ONLY bridge between the Python facade and the compute path.  There is no Python or
CPU fallback: if the library is missing this module raises ImportError, and every batched call
fails with RuntimeError when the HIP runtime reports an error (e.g. no GPU).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int8, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p, POINTER

# (GPU_MAX_HW_QUEUES: the HIP runtime maps a process's streams onto 4 hardware queues by default and the engine uses up to 7 streams.  8
# queues help a process that runs only the continuous-batching solver legs (+6 %) and HALVE the two-kernel legs inside a full bench.py
# run, where the streams of earlier legs have moved the mapping - INTEGRATION.md section 6, profiles/r6/hardware_queues_ab.json.  Not set here.)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAZ_LIB_PATH") or os.path.join(_HERE, "csrc", "libraz.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built. "
        "Run `python -c \"import __graft_entry__ as g; g.build()\"` at the repo root "
        "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")

try:
    # PyTorch (the HBM allocator and stream provider of every device entry point) ships its own copy of the HIP
    # runtime.  Load it FIRST: libraz.so then binds to that same libamdhip64; loaded in the other order the process
    # holds two runtimes that do not see each other's allocations ("no ROCm-capable device is detected").
    import torch  # noqa: F401
except Exception:  # host-only use (scalar bitboard entry points, row emission) works without it
    pass
lib = ctypes.CDLL(LIB_PATH)

u64p = POINTER(c_uint64)
u8p = POINTER(c_uint8)

# name -> (restype, argtypes).  Device-pointer arguments are passed as integers (c_void_p).
SIGNATURES = {
    "raz_abi_version": (c_int, []),
    "raz_last_error": (c_char_p, []),
    "raz_find_correct_moves": (c_uint64, [c_uint64, c_uint64]),
    "raz_calc_flip": (c_uint64, [c_int, c_uint64, c_uint64]),
    "raz_bit_count": (c_int, [c_uint64]),
    "raz_flip_vertical": (c_uint64, [c_uint64]),
    "raz_flip_diag_a1h8": (c_uint64, [c_uint64]),
    "raz_rotate90": (c_uint64, [c_uint64]),
    "raz_rotate180": (c_uint64, [c_uint64]),
    "raz_bit_to_array": (c_int, [c_uint64, c_int, c_void_p]),
    "raz_env_step": (c_int, [u64p, u64p, u8p, u8p, u64p, c_int]),
    "raz_legal_moves_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "raz_calc_flip_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "raz_step_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "raz_sweep_forms": (c_int, [c_size_t, c_void_p, c_void_p]),
    "raz_score_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "raz_d4_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "raz_planes_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "raz_pick_kth_legal_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}



class RazNet(ctypes.Structure):
    """raz_net (include/raz.h)."""
    _fields_ = [("filters", ctypes.c_int32), ("res_layers", ctypes.c_int32), ("value_fc", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("d_weights", c_void_p), ("weight_bytes", c_size_t)]


class RazEngineConfig(ctypes.Structure):
    """raz_engine_config (include/raz.h)."""
    _fields_ = [("thinking_loop", ctypes.c_int32), ("required_visit_to_decide_action", ctypes.c_int32),
                ("start_rethinking_turn", ctypes.c_int32), ("change_tau_turn", ctypes.c_int32),
                ("virtual_loss", ctypes.c_int32), ("allowed_resign_turn", ctypes.c_int32),
                ("has_resign_threshold", ctypes.c_int32), ("share_mtcs_info", ctypes.c_int32),
                ("mirror_updates", ctypes.c_int32), ("record_root_w", ctypes.c_int32),
                ("c_puct", ctypes.c_double), ("noise_eps", ctypes.c_double), ("dirichlet_alpha", ctypes.c_double),
                ("resign_threshold", ctypes.c_double), ("disable_resignation_rate", ctypes.c_double),
                ("n_games", c_uint32), ("nodes_per_game", c_uint32), ("table_slots", c_uint32),
                ("max_plies", c_uint32), ("seed", c_uint32), ("reserved", c_uint32),
                ("use_solver_turn", ctypes.c_int32), ("use_solver_turn_in_simulation", ctypes.c_int32),
                ("solver_memo_slots", c_uint32), ("parallel_search_num", c_uint32), ("pool_bytes_per_game", c_uint64),
                ("solver_pool_waves", c_uint32), ("reserved2", c_uint32)]


class RazHarvestResult(ctypes.Structure):
    _fields_ = [("harvested", c_uint32), ("restarted", c_uint32), ("skipped", c_uint32), ("playing", c_uint32)]


class RazEngineStats(ctypes.Structure):
    _fields_ = [("finished_games", c_uint64), ("total_sims", c_uint64), ("nn_leaves", c_uint64),
                ("error_flags", c_uint64), ("selections", c_uint64), ("max_pool_used", c_uint64), ("idle_or_done", c_uint64),
                ("max_pool_bytes", c_uint64)]


SIGNATURES.update({
    "raz_net_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "raz_net_scratch_bytes": (c_size_t, [c_int, c_int, c_size_t]),
    "raz_net_load": (c_int, [POINTER(RazNet), ctypes.c_char_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "raz_net_forward": (c_int, [POINTER(RazNet), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                c_void_p, c_size_t, c_void_p]),
    "raz_net_range_check": (c_int, [POINTER(RazNet), POINTER(c_int), c_void_p]),
    "raz_net_range_stats": (c_int, [POINTER(RazNet), POINTER(c_int), POINTER(ctypes.c_ulonglong), c_void_p]),
    "raz_engine_workspace_bytes": (c_size_t, [POINTER(RazEngineConfig)]),
    "raz_engine_create": (c_int, [POINTER(RazEngineConfig), POINTER(RazNet), c_void_p, c_size_t, c_void_p,
                                  c_size_t, POINTER(c_void_p)]),
    "raz_engine_destroy": (None, [c_void_p]),
    "raz_engine_start": (c_int, [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p]),
    "raz_emit_game_rows_json": (ctypes.c_longlong, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                                    POINTER(c_int)]),
    "raz_format_float_repr": (c_int, [ctypes.c_double, ctypes.c_char_p]),
    "raz_engine_next_game": (c_int, [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p]),
    "raz_engine_step": (c_int, [c_void_p, c_uint32, c_void_p]),
    "raz_engine_step_timed": (c_int, [c_void_p, c_uint32, POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_void_p]),
    "raz_engine_set_position": (c_int, [c_void_p, c_uint32, c_uint64, c_uint64, c_int, c_uint32, c_int, c_int, c_void_p]),
    "raz_engine_set_positions": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_uint32, c_int, c_int, c_void_p]),
    "raz_engine_read_node": (c_int, [c_void_p, c_uint32, c_uint64, c_uint64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     POINTER(c_int), c_void_p]),
    "raz_engine_stop_thinking": (c_int, [c_void_p, c_uint32, c_void_p]),
    "raz_engine_adopt_tree": (c_int, [c_void_p, c_uint32, c_int, c_void_p]),
    "raz_engine_gc": (c_int, [c_void_p, c_uint32, c_void_p]),
    "raz_engine_set_parts": (c_int, [c_void_p, c_int]),
    "raz_engine_set_solver_pool_every": (c_int, [c_void_p, c_int]),
    "raz_engine_stats_sync": (c_int, [c_void_p, POINTER(RazEngineStats), c_void_p]),
    "raz_engine_read_records": (c_int, [c_void_p] * 11 + [c_void_p]),
    "raz_engine_device_ptr": (c_void_p, [c_void_p, c_int]),
    "raz_engine_records_extent": (c_int, [c_void_p, c_uint32, c_uint32, POINTER(c_uint32), c_void_p]),
    "raz_engine_pack_records": (c_int, [c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "raz_engine_harvest": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p,
                                   c_void_p, POINTER(RazHarvestResult), c_void_p]),
    "raz_leaf_cache_bytes": (c_size_t, [c_uint32, c_size_t]),
    "raz_engine_set_leaf_cache": (c_int, [c_void_p, c_void_p, c_size_t, c_uint32, c_uint32, c_void_p]),
    "raz_engine_leaf_cache_stats": (c_int, [c_void_p, c_void_p, c_void_p]),
    "raz_engine_solver_stats": (c_int, [c_void_p, c_void_p, c_void_p]),
    "raz_engine_debug_read": (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_void_p]),
    "raz_engine_set_resign_threshold": (c_int, [c_void_p, c_int, ctypes.c_double]),
})

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == ABI mismatch: fail at import, loudly
    _fn.restype = _res
    _fn.argtypes = _args

if lib.raz_abi_version() != 3:
    raise ImportError(f"libraz ABI version {lib.raz_abi_version()} != 3 (stale build?)")


def last_error() -> str:
    return (lib.raz_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "libraz"):
    """Raise RuntimeError on a negative status code (SURVEY §8(b) error convention)."""
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def tensor_ptr(t):
    """Device pointer of a contiguous torch tensor (torch is only the memory allocator here)."""
    if not t.is_contiguous():
        raise ValueError("libraz needs contiguous tensors")
    return t.data_ptr()


def current_stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
