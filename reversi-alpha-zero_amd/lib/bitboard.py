"""Drop-in for reversi_zero/lib/bitboard.py, backed by libraz (C ABI, include/raz.h).

Same names, argument meaning and error behaviour as the reference (lib/bitboard.py:9-171):
bitboards are Python ints < 2**64, bit 0 = top-left.  Scalar calls run the host instantiation of
the library's primitives; the `*_batch` functions launch the gfx950 sweep kernels on device
tensors (torch.int64 tensors are reinterpreted as u64; they are plumbing for HBM pointers only).
"""
import numpy as np

from .._native import lib, check, tensor_ptr, current_stream_ptr

BLACK_CHR = "O"
WHITE_CHR = "X"
EXTRA_CHR = "*"

_M64 = 0xFFFFFFFFFFFFFFFF


def board_to_string(black, white, with_edge=True, extra=None):
    """ASCII rendering, lib/bitboard.py:9-50 (pure formatting: stays in Python)."""
    extra = extra or 0
    cells = []
    for i in range(64):
        bit = 1 << i
        cells.append(BLACK_CHR if black & bit else WHITE_CHR if white & bit else
                     EXTRA_CHR if extra & bit else " ")
    border = "#" * 10 + "\n" if with_edge else ""
    side = "#" if with_edge else ""
    rows = [side + "".join(cells[y * 8:y * 8 + 8]) + side + "\n" for y in range(8)]
    return border + "".join(rows) + border


def find_correct_moves(own, enemy):
    """Legal-move mask (lib/bitboard.py:53-67)."""
    return lib.raz_find_correct_moves(own & _M64, enemy & _M64)


def calc_flip(pos, own, enemy):
    """Flip mask when `own` plays `pos` (lib/bitboard.py:70-81; asserts 0 <= pos <= 63 like :78)."""
    assert 0 <= pos <= 63, f"pos={pos}"
    return lib.raz_calc_flip(int(pos), own & _M64, enemy & _M64)


def bit_count(x):
    """lib/bitboard.py:132-133."""
    return lib.raz_bit_count(x & _M64)


def bit_to_array(x, size):
    """bit_to_array(0b0010, 4) -> array([0, 1, 0, 0], dtype=uint8) (lib/bitboard.py:136-138)."""
    out = np.zeros(max(size, 0), dtype=np.uint8)
    n = min(size, 64)
    if n > 0:
        check(lib.raz_bit_to_array(x & _M64, n, out.ctypes.data), "raz_bit_to_array")
    return out


def flip_vertical(x):
    return lib.raz_flip_vertical(x & _M64)


def flip_diag_a1h8(x):
    return lib.raz_flip_diag_a1h8(x & _M64)


def rotate90(x):
    """Rotate the board RIGHT once (lib/bitboard.py:154-155)."""
    return lib.raz_rotate90(x & _M64)


def rotate180(x):
    return lib.raz_rotate180(x & _M64)


def b64(x):
    return x & _M64


def dirichlet_noise_of_mask(mask, alpha):
    """Dirichlet noise scattered onto the set bits of `mask` in ascending bit order
    (lib/bitboard.py:162-171).  Host convenience for the facade; the engine draws its root noise
    on device from its own counter-based stream."""
    num_1 = bit_count(mask)
    noise = list(np.random.dirichlet([alpha] * num_1))
    ret = np.zeros(64)
    for i in range(64):
        if (1 << i) & mask:
            ret[i] = noise.pop(0)
    return ret


# ---- batched device forms (no reference counterpart: the reference is one position at a time) ----

def _u64(t):
    import torch
    if t.dtype not in (torch.int64, torch.uint64):
        raise TypeError("bitboard tensors must be int64/uint64")
    if not t.is_cuda:
        raise ValueError("batched bitboard ops are device-only (no CPU fallback)")
    return tensor_ptr(t)


def legal_moves_batch(own, enemy, out=None):
    import torch
    out = torch.empty_like(own) if out is None else out
    check(lib.raz_legal_moves_batch(_u64(own), _u64(enemy), _u64(out), own.numel(),
                                    current_stream_ptr()), "raz_legal_moves_batch")
    return out


def calc_flip_batch(pos, own, enemy, out=None):
    import torch
    out = torch.empty_like(own) if out is None else out
    check(lib.raz_calc_flip_batch(tensor_ptr(pos), _u64(own), _u64(enemy), _u64(out), own.numel(),
                                  current_stream_ptr()), "raz_calc_flip_batch")
    return out


def step_batch(black, white, player, status, legal, action):
    """In-place ReversiEnv.step over a batch (see include/raz.h raz_step_batch)."""
    check(lib.raz_step_batch(_u64(black), _u64(white), tensor_ptr(player), tensor_ptr(status),
                             _u64(legal), tensor_ptr(action), black.numel(), current_stream_ptr()),
          "raz_step_batch")


def sweep_forms(n):
    """(legal_moves_form, step_form) of a batch of n boards: which kernels its whole superblocks run on (include/raz.h raz_sweep_forms:
    0 = a board per lane, 1 = bit-sliced, step 2 = the hybrid form)."""
    import ctypes
    a, b = ctypes.c_int(-1), ctypes.c_int(-1)
    check(lib.raz_sweep_forms(int(n), ctypes.addressof(a), ctypes.addressof(b)), "raz_sweep_forms")
    return a.value, b.value


def score_batch(black, white):
    import torch
    winner = torch.empty(black.numel(), dtype=torch.uint8, device=black.device)
    diff = torch.empty(black.numel(), dtype=torch.int8, device=black.device)
    check(lib.raz_score_batch(_u64(black), _u64(white), tensor_ptr(winner), tensor_ptr(diff),
                              black.numel(), current_stream_ptr()), "raz_score_batch")
    return winner, diff


def d4_batch(x, sym, out=None):
    import torch
    out = torch.empty_like(x) if out is None else out
    check(lib.raz_d4_batch(_u64(x), _u64(out), tensor_ptr(sym), x.numel(), current_stream_ptr()),
          "raz_d4_batch")
    return out


def planes_batch(own, enemy):
    import torch
    out = torch.empty((own.numel(), 2, 8, 8), dtype=torch.float32, device=own.device)
    check(lib.raz_planes_batch(_u64(own), _u64(enemy), tensor_ptr(out), own.numel(),
                               current_stream_ptr()), "raz_planes_batch")
    return out


def pick_kth_legal_batch(legal, rnd):
    import torch
    out = torch.empty(legal.numel(), dtype=torch.uint8, device=legal.device)
    check(lib.raz_pick_kth_legal_batch(_u64(legal), tensor_ptr(rnd), tensor_ptr(out), legal.numel(),
                                       current_stream_ptr()), "raz_pick_kth_legal_batch")
    return out
