"""CPU: the NET kernels on EMULATED matrix cores against the CPU oracle.  The exact-f32 family (raznet-forward-v1), bit for bit:
k_net_mfma (v_mfma_f32_16x16x4_f32, narrow nets), k_net_wave (VALU + LDS, any shape), k_conv0_wide / k_conv3x3_wide / k_heads_wide
(v_mfma_f32_32x32x2_f32, F >= 128) - csrc/raz_net.hip, raz_net_mfma.hip, raz_net_wide.hip compiled for the host against the wave
emulator (tests/native/wave_emu: fibers for lanes, a matrix-core instruction = an all-gather of the operands + the k-ordered
fmaf chains of its documented lane layout).  The GPU tests (tests/test_engine_gpu.py) are the tests of record; this is the loop in
which a net kernel's indexing can be developed without a GPU - the emulation is self-validating: a wrong operand layout or
accumulation order does not reproduce the oracle.  The split-f16 trunk (raznet-forward-v2, csrc/raz_net_f16x3.hip: v_mfma_f32_32x32x16_f16
with hi/lo operand pairs, LDS-DMA staging) runs here too, compared at its stated tolerance: the matrix core's internal summation
order is not documented, so the emulation fixes one (k ascending) and checks layout, staging and the split arithmetic, not bits."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as O
from conftest import ROOT, _locked

EMU_DIR = os.path.join(ROOT, "tests", "native", "wave_emu")
LIB = os.path.join(ROOT, "tests", "native", "libraz_emu_net.so")


@pytest.fixture(scope="module")
def lib():
    with _locked("emu"):
        r = subprocess.run(["make", "-C", EMU_DIR, "../libraz_emu_net.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    from reversi_alpha_zero_amd import _native as N
    lib = ctypes.CDLL(LIB)
    lib.raz_last_error.restype = ctypes.c_char_p
    for name in ("raz_net_weight_bytes", "raz_net_scratch_bytes", "raz_net_load", "raz_net_forward", "raz_net_range_stats"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = N.SIGNATURES[name]
    return lib


def _forward(lib, blob, own, enemy, reserved=0, active=None, stats=None):
    from reversi_alpha_zero_amd import _native as N
    _, _, F, R, V = struct.unpack_from("<5i", blob, 0)
    w = np.zeros(lib.raz_net_weight_bytes(F, R, V), dtype=np.uint8)
    net = N.RazNet()
    net.reserved = reserved
    assert lib.raz_net_load(ctypes.byref(net), blob, len(blob), w.ctypes.data, w.size, None) == 0, lib.raz_last_error()
    n = len(own)
    need = lib.raz_net_scratch_bytes(F, V, n)
    scratch = np.zeros(max(need, 8), dtype=np.uint8)
    pol, val = np.full((n, 64), 7.0, np.float32), np.full(n, 7.0, np.float32)
    rc = lib.raz_net_forward(ctypes.byref(net), own.ctypes.data, enemy.ctypes.data, active.ctypes.data if active is not None else None,
                             pol.ctypes.data, val.ctypes.data, n, scratch.ctypes.data if need else None, need, None)
    assert rc == 0, lib.raz_last_error()
    if stats is not None:
        over, rows = ctypes.c_int(0), ctypes.c_ulonglong(0)
        assert lib.raz_net_range_stats(ctypes.byref(net), ctypes.byref(over), ctypes.byref(rows), None) == 0, lib.raz_last_error()
        stats.update(overflowed=bool(over.value), rows_repaired=int(rows.value))
    return pol, val


def _oracle(blob, own, enemy):
    o = O.load_ext()
    pol, val = np.zeros((len(own), 64), np.float32), np.zeros(len(own), np.float32)
    for i in range(len(own)):
        v = np.zeros(1, np.float32)
        assert o.orc_net_forward(blob, len(blob), int(own[i]), int(enemy[i]), pol[i].ctypes.data, v.ctypes.data) == 0
        val[i] = v[0]
    return pol, val


def _positions(n, seed):
    rng = np.random.default_rng(seed)
    own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    return own, rng.integers(0, 2**64, size=n, dtype=np.uint64) & ~own


@pytest.mark.parametrize("shape,reserved", [((16, 1, 16), 0), ((16, 2, 48), 0), ((32, 1, 32), 0), ((16, 1, 16), 1), ((48, 1, 20), 0),
                                            ((16, 1, 64), 0), ((16, 1, 80), 0)])
def test_emulated_narrow_net_kernels_equal_oracle(lib, shape, reserved, monkeypatch):
    """k_net_mfma (reserved 0, F in {16, 32}: 16x16x4 matrix-core tiles; R = 1 with the weights hoisted into registers, R = 2 without),
    k_net_wave (reserved 1, and any shape the matrix-core kernel does not take: F = 48) == the oracle's C net, with an active mask (skipped rows stay untouched).
    The grid is capped at 3 workgroups so that every workgroup loops over several positions (the registers that stay resident
    across positions, the prefetched dense-head operands: value widths 16..64 take that path, 80 the plain one)."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    monkeypatch.setenv("RAZ_NET_MAXGRID", "3")
    blob = ReversiNet(*shape).keras_init_(3).randomize_bn_(4).to_blob()
    n = 11
    own, enemy = _positions(n, 5)
    active = (np.arange(n) % 4 != 2).astype(np.uint8)
    pol, val = _forward(lib, blob, own, enemy, reserved, active)
    rp, rv = _oracle(blob, own, enemy)
    on = active.astype(bool)
    assert np.array_equal(pol[on].view(np.uint32), rp[on].view(np.uint32)) and np.array_equal(val[on].view(np.uint32), rv[on].view(np.uint32))
    assert (pol[~on] == 7.0).all() and (val[~on] == 7.0).all()


def test_emulated_wide_net_kernels_equal_oracle(lib):
    """k_conv0_wide + k_conv3x3_wide (implicit GEMM on 32x32x2 matrix-core tiles, 4-wave workgroups, chunk-major K order) +
    k_heads_wide on a 128-filter net == the oracle, bit for bit, on a batch that is not a multiple of the 4 positions per workgroup."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    blob = ReversiNet(128, 1, 32).keras_init_(6).randomize_bn_(7).to_blob()
    own, enemy = _positions(6, 8)
    pol, val = _forward(lib, blob, own, enemy, 0)
    rp, rv = _oracle(blob, own, enemy)
    assert np.array_equal(pol.view(np.uint32), rp.view(np.uint32)) and np.array_equal(val.view(np.uint32), rv.view(np.uint32))


@pytest.mark.parametrize("shape,n", [((128, 1, 32), 11), ((256, 1, 16), 3)])
def test_emulated_split_f16_trunk_is_within_tolerance_of_the_oracle(lib, shape, n):
    """raznet-forward-v2 (reserved 4): k_conv0_split + k_conv3x3_f16x3 (8-wave workgroups = 8 positions x 128 output channels, 48-stage
    LDS-DMA pipeline, three f16 matrix instructions per product) + k_heads_split within 1e-5 of the oracle's f32 net (the tolerance
    include/raz.h states for this path), on a ragged batch (F = 128: a second, partly filled position group; F = 256: two output-channel
    tiles per group) with an active mask; skipped rows stay untouched and the range flag stays clear."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    blob = ReversiNet(*shape).keras_init_(6).randomize_bn_(7).to_blob()
    own, enemy = _positions(n, 5)
    active = (np.arange(n) % 5 != 3).astype(np.uint8)
    pol, val = _forward(lib, blob, own, enemy, 4, active)
    rp, rv = _oracle(blob, own, enemy)
    on = active.astype(bool)
    assert np.abs(pol[on] - rp[on]).max() <= 1e-5 and np.abs(val[on] - rv[on]).max() <= 1e-5
    assert (pol[~on] == 7.0).all() and (val[~on] == 7.0).all()


def _net_that_overflows_on_crowded_boards(F, V):
    """Stem weights all equal, so that a stem activation is 10^4 x the discs in the square's 3x3 neighbourhood: boards with at most 3
    discs in any neighbourhood stay below the f16 range, crowded boards leave it (9 discs: 9 x 10^4 > 60000).  The residual convs
    are scaled down so that the trunk output stays the stem's."""
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    net = ReversiNet(F, 1, V).keras_init_(6)
    with torch.no_grad():
        net.stem.conv.weight.fill_(1.0e4)
        net.stem.conv.bias.zero_()
        for cb in net.res[0]:
            cb.conv.weight.mul_(1.0e-7)
    return net.to_blob()


@pytest.mark.parametrize("shape,n_crowded", [((128, 1, 32), 40), ((256, 1, 16), 2)])
def test_emulated_rows_that_leave_the_f16_range_are_evaluated_by_the_exact_f32_chains(lib, shape, n_crowded):
    """raznet-forward-v2's range repair (csrc/raz_net.hip k_net_wave_repair): in a batch mixing sparse boards (in range) and crowded
    boards (stem activations beyond 60000), the crowded rows come out as raznet-forward-v1 evaluates them - the oracle, bit for bit -
    and the sparse rows exactly as a v2 forward of the sparse rows alone computes them: a row's answer depends on its position only.
    Up to 32 repaired rows per forward leave the sticky flag down; more raise it (the caller then moves to the f32 kernels)."""
    F, _, V = shape
    blob = _net_that_overflows_on_crowded_boards(F, V)
    rng = np.random.default_rng(11)
    n_sparse = 5
    sparse_own = np.array([1 << int(rng.integers(0, 64)) for _ in range(n_sparse)], dtype=np.uint64)
    sparse_enemy = np.array([(1 << int(rng.integers(0, 64))) & ~int(o) for o in sparse_own], dtype=np.uint64)
    crowded_own = rng.integers(0, 2**64, size=n_crowded, dtype=np.uint64)
    crowded_enemy = ~crowded_own   # every square taken
    order = rng.permutation(n_sparse + n_crowded)
    own = np.concatenate([sparse_own, crowded_own])[order]
    enemy = np.concatenate([sparse_enemy, crowded_enemy])[order]
    crowded = (order >= n_sparse)
    st = {}
    pol, val = _forward(lib, blob, own, enemy, 4, stats=st)
    rp, rv = _oracle(blob, own, enemy)
    assert np.isfinite(pol).all() and np.isfinite(val).all()
    assert np.array_equal(pol[crowded].view(np.uint32), rp[crowded].view(np.uint32)) and np.array_equal(val[crowded].view(np.uint32), rv[crowded].view(np.uint32))
    st2 = {}
    sp, sv = _forward(lib, blob, own[~crowded], enemy[~crowded], 4, stats=st2)
    assert np.array_equal(pol[~crowded].view(np.uint32), sp.view(np.uint32)) and np.array_equal(val[~crowded].view(np.uint32), sv.view(np.uint32))
    assert st2 == {"overflowed": False, "rows_repaired": 0}
    assert st == {"overflowed": n_crowded > 32, "rows_repaired": n_crowded}
