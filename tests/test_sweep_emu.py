"""CPU: the batched sweep kernels of csrc/raz_sweep.hip (k_legal_moves, k_calc_flip, k_step, k_score, k_d4, k_pick_kth) run by
the wave emulator (tests/native/wave_emu) against the CPU oracle: what the GPU tests (tests/test_sweep_gpu.py, the tests of
record) check on the device, here for the kernels' LOGIC - the 256-board wave blocks and the ragged tails, the out-of-line round
for boards whose opponent cannot move, finished games left untouched - on sizes a CPU finishes in seconds.  The host build takes
the plain-C++ branch of the VALU-shaped primitives (their device forms are checked by tests/native/bbv_check.cpp + the GPU tests)."""
import ctypes
import os

import subprocess
import sys

# (read once by the library) every whole superblock of 2048 boards runs on the bit-sliced kernels: k_legal_moves_sliced, and for
# ReversiEnv.step the form the library adopts for large batches, k_step_hybrid - by two waves, so that a wave takes several superblocks;
# k_step_sliced (RAZ_SWEEP_SLICED_STEP=1) runs in a process of its own (test_emulated_everything_sliced_step_form_...)
_EMU_ENV = {"RAZ_SWEEP_HYBRID_WAVES": "2", "RAZ_SWEEP_SLICED_MIN": "2048"}

import numpy as np
import pytest

import oracle as O
from emu_util import load


@pytest.fixture(scope="module", autouse=True)
def _every_superblock_sliced():
    """Set while THIS module's tests run - not at import: collecting the file in a run of the GPU tests must not change what the device
    library does (the emulator library reads the thresholds at its first sweep call, which is in here)."""
    old = {k: os.environ.get(k) for k in _EMU_ENV}
    for k, v in _EMU_ENV.items():
        os.environ.setdefault(k, v)
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _ptr(a):
    return a.ctypes.data


def _playout_positions(n, seed):
    """Positions at random plies of random playouts (oracle env), with a legal action each - incl. late positions where a move
    leaves the opponent without a move (pass) or ends the game."""
    orc = O.load()
    rng = np.random.default_rng(seed)
    black, white, player = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
    action = np.zeros(n, np.uint8)
    env = O.OrcEnv()
    for i in range(n):
        orc.orc_env_reset(ctypes.byref(env))
        for _ in range(64 if i % 2 == 0 else int(rng.integers(0, 60))):   # every other game: up to its last live position
            o, e = (env.black, env.white) if env.next_player == 1 else (env.white, env.black)
            legal = orc.orc_find_correct_moves(o, e)
            moves = [s for s in range(64) if legal >> s & 1]
            prev = (env.black, env.white, env.next_player)
            orc.orc_env_step(ctypes.byref(env), int(moves[rng.integers(0, len(moves))]))
            if env.done:
                orc.orc_env_update(ctypes.byref(env), *prev)
                break
        black[i], white[i], player[i] = env.black, env.white, env.next_player
        o, e = (env.black, env.white) if env.next_player == 1 else (env.white, env.black)
        legal = orc.orc_find_correct_moves(o, e)
        moves = [s for s in range(64) if legal >> s & 1]
        action[i] = moves[rng.integers(0, len(moves))]
    return black, white, player, action


@pytest.mark.parametrize("n", [0, 1, 3, 255, 256, 257, 1030, 2048, 4096 + 333])
def test_emulated_legal_moves_and_flips_equal_oracle(n):
    """(n >= 2048: the whole superblocks on the bit-sliced kernel of csrc/raz_sweep_sliced.h, the rest a board per lane.)"""
    lib = load()
    rng = np.random.default_rng(n + 5)
    own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    enemy = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    enemy[: n // 2] &= ~own[: n // 2]            # half playable-looking, half overlapping garbage
    pos = rng.integers(0, 70, size=n, dtype=np.uint8)   # incl. out-of-range (> 63 -> 0)
    legal, flip = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    assert lib.raz_legal_moves_batch(_ptr(own), _ptr(enemy), _ptr(legal), n, None) == 0
    assert lib.raz_calc_flip_batch(_ptr(pos), _ptr(own), _ptr(enemy), _ptr(flip), n, None) == 0
    assert np.array_equal(legal, O.np_find_correct_moves(own, enemy))
    ref = O.np_calc_flip(np.minimum(pos, 63), own, enemy)
    ref[pos > 63] = 0
    assert np.array_equal(flip, ref)


@pytest.mark.parametrize("n", [5, 256, 777, 2048 + 300, 3 * 2048 + 17])
def test_emulated_step_equals_oracle_incl_passes_and_finished_games(n):
    """(n >= 2048: the first superblock on the bit-sliced kernel.)"""
    lib = load()
    black, white, player, action = _playout_positions(n, n)
    status = np.zeros(n, np.uint8)
    status[::11] = 1                         # finished games are left untouched (legal = 0)
    action[3::17] = 255                      # resignations
    b, w, p, s = black.copy(), white.copy(), player.copy(), status.copy()
    legal = np.zeros(n, np.uint64)
    if n >= 2048:   # what is tested: the sliced forms (the library's own answer), on arrays k_step_hybrid accepts
        lf, sf = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib.raz_sweep_forms(n, ctypes.addressof(lf), ctypes.addressof(sf)) == 0
        assert (lf.value, sf.value) == (1, int(os.environ.get("RAZ_SWEEP_SLICED_STEP", "2"))), (lf.value, sf.value)
        assert (_ptr(p) | _ptr(s) | _ptr(action)) % 16 == 0
    assert lib.raz_step_batch(_ptr(b), _ptr(w), _ptr(p), _ptr(s), _ptr(legal), _ptr(action), n, None) == 0
    ob, ow, op, os_, ol = O.np_step(black, white, player, status, action)
    assert np.array_equal(b, ob) and np.array_equal(w, ow) and np.array_equal(p, op) and np.array_equal(s, os_) and np.array_equal(legal, ol)
    passed = ((os_ == 0) & (op == player) & (status == 0) & (action < 64)).sum()      # the opponent had no move: the mover moves again
    ended = ((os_ != 0) & (status == 0) & (action < 64)).sum()                        # neither side had one: counted and over
    assert n < 100 or passed + ended > 0   # the out-of-line second mobility round of k_step was exercised


def test_emulated_score_d4_and_pick_equal_host_primitives():
    lib = load()
    n = 515
    rng = np.random.default_rng(9)
    black = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    white = rng.integers(0, 2**64, size=n, dtype=np.uint64) & ~black
    winner, diff = np.zeros(n, np.uint8), np.zeros(n, np.int8)
    assert lib.raz_score_batch(_ptr(black), _ptr(white), _ptr(winner), _ptr(diff), n, None) == 0
    pc = lambda a: np.unpackbits(a.view(np.uint8)).reshape(n, 64).sum(1).astype(np.int64)
    d = pc(black) - pc(white)
    assert np.array_equal(diff.astype(np.int64), d) and np.array_equal(winner, np.where(d > 0, 1, np.where(d < 0, 2, 3)).astype(np.uint8))
    sym = rng.integers(0, 8, size=n, dtype=np.uint8)
    out = np.zeros(n, np.uint64)
    assert lib.raz_d4_batch(_ptr(black), _ptr(out), _ptr(sym), n, None) == 0
    orc = O.load()
    for i in range(n):
        x = int(black[i])
        if sym[i] >> 2:
            x = orc.orc_flip_vertical(x)
        for _ in range(int(sym[i]) & 3):
            x = orc.orc_rotate90(x)
        assert int(out[i]) == x, i
    legal = black | (np.uint64(1) << np.uint64(7))
    rnd = rng.integers(0, 2**31 - 1, size=n, dtype=np.uint32)
    act = np.zeros(n, np.uint8)
    assert lib.raz_pick_kth_legal_batch(_ptr(legal), _ptr(rnd), _ptr(act), n, None) == 0
    for i in range(n):
        bits = [s for s in range(64) if int(legal[i]) >> s & 1]
        assert act[i] == bits[int(rnd[i]) % len(bits)]


def test_emulated_sliced_step_on_garbage_boards_and_every_kind_of_action():
    """The bit-sliced step kernel (csrc/raz_sweep_sliced.h) on 2048 + 2048 boards of overlapping own / enemy garbage, actions on occupied
    squares, no-flip moves, resignations, actions outside the board (64..254: treated as a move that flips nothing), finished games:
    == the board-per-lane kernel's primitives, i.e. the oracle's step."""
    lib = load()
    n = 4096
    rng = np.random.default_rng(77)
    black = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    white = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    white[: n // 2] &= ~black[: n // 2]
    thin = rng.integers(0, 2**64, size=n, dtype=np.uint64) & rng.integers(0, 2**64, size=n, dtype=np.uint64)
    black[n // 4: n // 2] &= thin[n // 4: n // 2]                # sparse boards: moves that flip something, passes, empty squares
    player = rng.integers(1, 3, size=n, dtype=np.uint8)
    status = np.where(rng.random(n) < 0.1, rng.integers(1, 4, size=n), 0).astype(np.uint8)
    action = rng.integers(0, 64, size=n, dtype=np.uint8)
    action[rng.random(n) < 0.05] = 255
    b, w, p, s = black.copy(), white.copy(), player.copy(), status.copy()
    legal = np.zeros(n, np.uint64)
    assert (_ptr(p) | _ptr(s) | _ptr(action)) % 16 == 0
    assert lib.raz_step_batch(_ptr(b), _ptr(w), _ptr(p), _ptr(s), _ptr(legal), _ptr(action), n, None) == 0
    ob, ow, op, os_, ol = O.np_step(black, white, player, status, action)
    for name, got, want in (("black", b, ob), ("white", w, ow), ("player", p, op), ("status", s, os_), ("legal", legal, ol)):
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (name, bad[:5], [hex(int(x)) for x in got[bad[:3]]], [hex(int(x)) for x in want[bad[:3]]])
    assert (os_ & 0x10).any() and (os_ & 0x20).any() and ((os_ == 0) & (status == 0)).sum() > n // 4


def test_emulated_everything_sliced_step_form_in_a_process_of_its_own():
    """k_step_sliced (RAZ_SWEEP_SLICED_STEP=1: the whole step bit-sliced, one wave per SIMD - kept beside k_step_hybrid, which the tests
    above run): the step tests with whole superblocks, in a child process (the library reads the form once)."""
    if os.environ.get("RAZ_SWEEP_TEST_CHILD") or os.environ.get("RAZ_SWEEP_SLICED_STEP"):
        pytest.skip("a child of this test, or a run with the form forced from outside")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-p", "no:xdist",
                        "-k", "(step_equals_oracle and 2348) or garbage"],
                       env={**os.environ, **_EMU_ENV, "RAZ_SWEEP_SLICED_STEP": "1", "RAZ_SWEEP_TEST_CHILD": "1"}, capture_output=True, text=True, timeout=1200,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]

