"""GPU: the batched bitboard sweep kernels (through the C ABI) against the CPU oracle — bit-exact —
on the golden vectors, on seeded random inputs, on ragged/empty sizes, and at full size through
size-independent game properties."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import H
import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64) if a.dtype == np.uint64 else np.ascontiguousarray(a)).to(DEV)


def to_np64(t):
    return t.cpu().numpy().view(np.uint64)


def harvest(golden_bb):
    own = np.array([H(r["own"]) for r in golden_bb["positions"]], dtype=np.uint64)
    enemy = np.array([H(r["enemy"]) for r in golden_bb["positions"]], dtype=np.uint64)
    return own, enemy


def test_legal_and_flip_golden(golden_bb):
    from reversi_alpha_zero_amd.lib import bitboard as bb
    recs = golden_bb["positions"]
    own, enemy = harvest(golden_bb)
    legal = to_np64(bb.legal_moves_batch(to_dev(own), to_dev(enemy)))
    assert [int(x) for x in legal] == [H(r["legal"]) for r in recs]
    po, pe, pp, pf = [], [], [], []
    for r in recs:
        for a, f in r["flips"].items():
            po.append(H(r["own"])); pe.append(H(r["enemy"])); pp.append(int(a)); pf.append(H(f))
    got = to_np64(bb.calc_flip_batch(to_dev(np.array(pp, dtype=np.uint8)), to_dev(np.array(po, dtype=np.uint64)),
                                     to_dev(np.array(pe, dtype=np.uint64))))
    assert [int(x) for x in got] == pf
    g = golden_bb["garbage"]
    own = np.array([H(r["own"]) for r in g], dtype=np.uint64)
    enemy = np.array([H(r["enemy"]) for r in g], dtype=np.uint64)
    pos = np.array([r["pos"] for r in g], dtype=np.uint8)
    assert [int(x) for x in to_np64(bb.legal_moves_batch(to_dev(own), to_dev(enemy)))] == [H(r["legal"]) for r in g]
    assert [int(x) for x in to_np64(bb.calc_flip_batch(to_dev(pos), to_dev(own), to_dev(enemy)))] == [H(r["flip"]) for r in g]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 257, 100003, 1 << 20])
def test_legal_flip_random_sizes_vs_oracle(n):
    from reversi_alpha_zero_amd.lib import bitboard as bb
    rng = np.random.default_rng(n + 1)
    own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    enemy = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    half = n // 2
    enemy[:half] &= ~own[:half]          # half legal-looking, half overlapping garbage
    pos = rng.integers(0, 70, size=n, dtype=np.uint8)  # includes out-of-range (>63) -> 0
    d_own, d_enemy, d_pos = to_dev(own), to_dev(enemy), to_dev(pos)
    legal = to_np64(bb.legal_moves_batch(d_own, d_enemy))
    flip = to_np64(bb.calc_flip_batch(d_pos, d_own, d_enemy))
    assert np.array_equal(legal, O.np_find_correct_moves(own, enemy))
    assert np.array_equal(flip, O.np_calc_flip(pos, own, enemy))


def test_step_golden_playouts_lockstep(golden_bb):
    """Replay the 40 reference playouts in lockstep; finished games idle (status != 0)."""
    from reversi_alpha_zero_amd.lib import bitboard as bb
    games = golden_bb["playouts"]
    n = len(games)
    T = max(len(g["actions"]) for g in games)
    black = torch.full((n,), 0x0000000810000000, dtype=torch.int64, device=DEV)
    white = torch.full((n,), 0x0000001008000000, dtype=torch.int64, device=DEV)
    player = torch.ones(n, dtype=torch.uint8, device=DEV)
    status = torch.zeros(n, dtype=torch.uint8, device=DEV)
    legal = torch.zeros(n, dtype=torch.int64, device=DEV)
    for t in range(T):
        act = np.array([g["actions"][t] if t < len(g["actions"]) else 0 for g in games], dtype=np.uint8)
        pl = player.cpu().numpy()
        for i, g in enumerate(games):
            if t < len(g["actions"]):
                assert pl[i] == g["players"][t]
        bb.step_batch(black, white, player, status, legal, to_dev(act))
    st = status.cpu().numpy()
    assert [int(x) for x in to_np64(black)] == [H(g["black"]) for g in games]
    assert [int(x) for x in to_np64(white)] == [H(g["white"]) for g in games]
    assert [int(s) for s in st] == [g["winner"] for g in games]
    winner, diff = bb.score_batch(black, white)
    assert [int(x) for x in winner.cpu()] == [g["winner"] for g in games]
    assert (legal == 0).all()


def test_step_edge_cases_vs_oracle():
    """resign (255), illegal no-flip moves, occupied squares, already-finished games, odd n."""
    from reversi_alpha_zero_amd.lib import bitboard as bb
    rng = np.random.default_rng(5)
    n = 4099
    black = np.full(n, 0x0000000810000000, dtype=np.uint64)
    white = np.full(n, 0x0000001008000000, dtype=np.uint64)
    player = rng.integers(1, 3, size=n, dtype=np.uint8)
    status = np.where(rng.random(n) < 0.1, rng.integers(1, 4, size=n), 0).astype(np.uint8)
    action = rng.integers(0, 64, size=n, dtype=np.uint8)
    action[rng.random(n) < 0.1] = 255
    d = [to_dev(x) for x in (black, white, player, status)]
    legal = torch.zeros(n, dtype=torch.int64, device=DEV)
    bb.step_batch(d[0], d[1], d[2], d[3], legal, to_dev(action))
    eb, ew, ep, es, el = O.np_step(black, white, player, status, action)
    assert np.array_equal(to_np64(d[0]), eb) and np.array_equal(to_np64(d[1]), ew)
    assert np.array_equal(d[2].cpu().numpy(), ep) and np.array_equal(d[3].cpu().numpy(), es)
    assert np.array_equal(to_np64(legal), el)
    assert (es & 0x10).any() and (es & 0x20).any()


def _random_playout(n, seed, check_every=None):
    from reversi_alpha_zero_amd.lib import bitboard as bb
    black = torch.full((n,), 0x0000000810000000, dtype=torch.int64, device=DEV)
    white = torch.full((n,), 0x0000001008000000, dtype=torch.int64, device=DEV)
    player = torch.ones(n, dtype=torch.uint8, device=DEV)
    status = torch.zeros(n, dtype=torch.uint8, device=DEV)
    legal = bb.legal_moves_batch(black, white)
    g = torch.Generator(device=DEV).manual_seed(seed)
    plies = 0
    while True:
        rnd = torch.randint(0, 2**31 - 1, (n,), generator=g, device=DEV, dtype=torch.int32)
        action = bb.pick_kth_legal_batch(legal, rnd)
        if check_every and plies % check_every == 0:
            before = [x.cpu().numpy().copy() for x in (black, white, player, status)]
            act_np = action.cpu().numpy()
        bb.step_batch(black, white, player, status, legal, action)
        if check_every and plies % check_every == 0:
            eb, ew, ep, es, el = O.np_step(before[0].view(np.uint64), before[1].view(np.uint64), before[2], before[3], act_np)
            assert np.array_equal(to_np64(black), eb) and np.array_equal(to_np64(white), ew)
            assert np.array_equal(player.cpu().numpy(), ep) and np.array_equal(status.cpu().numpy(), es)
            assert np.array_equal(to_np64(legal), el)
        plies += 1
        if plies >= 60 and bool((status != 0).all()):
            break
        assert plies < 130
    return black, white, player, status, legal, plies


def test_random_playouts_every_ply_vs_oracle():
    _random_playout(20011, seed=3, check_every=1)


def test_full_size_playout_properties():
    """2^22 concurrent games to the end: size-independent invariants of Reversi."""
    from reversi_alpha_zero_amd.lib import bitboard as bb
    n = 1 << 22
    black, white, player, status, legal, plies = _random_playout(n, seed=11)
    b, w = to_np64(black), to_np64(white)
    assert not (b & w).any()                                   # colours disjoint
    st = status.cpu().numpy()
    assert ((st >= 1) & (st <= 3)).all()                       # natural endings only
    assert not O.np_find_correct_moves(b, w).any() and not O.np_find_correct_moves(w, b).any()
    winner, diff = bb.score_batch(black, white)
    assert np.array_equal(winner.cpu().numpy(), st)
    pc = lambda x: np.array([bin(int(v)).count("1") for v in x[:50000]])
    d = diff.cpu().numpy()[:50000]
    assert np.array_equal(pc(b) - pc(w), d)
    assert np.array_equal(np.sign(d), np.select([st[:50000] == 1, st[:50000] == 2], [1, -1], 0))
    total = pc(b) + pc(w)
    assert total.max() == 64 and total.min() >= 9 and 0.9 < (total == 64).mean() <= 1.0


def test_d4_and_planes_vs_oracle(orc, golden_bb):
    from reversi_alpha_zero_amd.lib import bitboard as bb
    rng = np.random.default_rng(9)
    n = 1001
    x = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    sym = rng.integers(0, 8, size=n, dtype=np.uint8)
    got = to_np64(bb.d4_batch(to_dev(x), to_dev(sym)))
    exp = []
    for v, s in zip(x, sym):
        v = int(v)
        if s & 4:
            v = orc.orc_flip_vertical(v)
        for _ in range(s & 3):
            v = orc.orc_rotate90(v)
        exp.append(v)
    assert [int(v) for v in got] == exp
    y = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    planes = bb.planes_batch(to_dev(x), to_dev(y)).cpu().numpy()
    assert planes.shape == (n, 2, 8, 8) and planes.dtype == np.float32
    bits = lambda a: ((a[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float32)
    assert np.array_equal(planes[:, 0].reshape(n, 64), bits(x))
    assert np.array_equal(planes[:, 1].reshape(n, 64), bits(y))


def test_which_kernel_forms_the_batches_run_on():
    """raz_sweep_forms: the library's own account of the forms - as adopted (board per lane below 2^25 boards, find_correct_moves
    sliced from there, ReversiEnv.step hybrid from 2^26), or, in a child of the test below, every whole superblock on the form forced."""
    from reversi_alpha_zero_amd.lib import bitboard as bb
    if os.environ.get("RAZ_SWEEP_TEST_CHILD"):
        assert bb.sweep_forms(2047) == (0, 0)
        assert bb.sweep_forms(2048) == (1, int(os.environ.get("RAZ_SWEEP_SLICED_STEP", "2")))
    elif not any(k.startswith("RAZ_SWEEP_") for k in os.environ):
        assert bb.sweep_forms(1 << 24) == (0, 0) and bb.sweep_forms((1 << 25) - 1) == (0, 0)
        assert bb.sweep_forms(1 << 25) == (1, 0) and bb.sweep_forms((1 << 26) - 1) == (1, 0)
        assert bb.sweep_forms(1 << 26) == (1, 2)


def test_batches_at_the_adopted_sizes_equal_the_board_per_lane_kernels_on_the_same_boards():
    """At BASELINE's largest sweep size and beyond, with the thresholds AS ADOPTED (no override): 2^26 + 2048 + 5 boards run
    find_correct_moves on k_legal_moves_sliced and ReversiEnv.step on k_step_hybrid (raz_sweep_forms says so); the same boards in pieces
    below 2^25 run on the board-per-lane kernels - every array must come out bit-identical.  Boards: disjoint random discs of every
    density, finished games, resignations, actions on occupied squares and on squares that flip nothing."""
    from reversi_alpha_zero_amd.lib import bitboard as bb
    if any(k.startswith("RAZ_SWEEP_") for k in os.environ):
        pytest.skip("thresholds forced from outside")
    n = (1 << 26) + 2048 + 5
    assert bb.sweep_forms(n) == (1, 2)
    g = torch.Generator(device=DEV).manual_seed(5)
    rnd = lambda: torch.randint(-(1 << 63), (1 << 63) - 1, (n,), dtype=torch.int64, device=DEV, generator=g)
    black = rnd()
    black[: n // 2] &= rnd()[: n // 2]                     # thinner boards in the first half, sparse ones in the first quarter
    black[: n // 4] &= rnd()[: n // 4]
    white = rnd() & ~black
    white[n // 8: n // 4] &= rnd()[n // 8: n // 4]
    player = torch.randint(1, 3, (n,), dtype=torch.uint8, device=DEV, generator=g)
    status = torch.where(torch.rand(n, device=DEV, generator=g) < 0.1, torch.randint(1, 4, (n,), dtype=torch.uint8, device=DEV, generator=g),
                         torch.zeros(n, dtype=torch.uint8, device=DEV))
    action = torch.randint(0, 64, (n,), dtype=torch.uint8, device=DEV, generator=g)
    action[torch.rand(n, device=DEV, generator=g) < 0.03] = 255   # (the contract: 0..63 or 255, include/raz.h raz_step_batch)
    own, enemy = torch.where(player == 1, black, white), torch.where(player == 1, white, black)
    piece = (1 << 24) + 1000                               # below every threshold: the board-per-lane kernels
    assert bb.sweep_forms(piece) == (0, 0)
    legal_big = bb.legal_moves_batch(own, enemy)
    for i in range(0, n, piece):
        assert torch.equal(legal_big[i:i + piece], bb.legal_moves_batch(own[i:i + piece].clone(), enemy[i:i + piece].clone()))
    del legal_big
    big = [t.clone() for t in (black, white, player, status)] + [torch.zeros(n, dtype=torch.int64, device=DEV)]
    bb.step_batch(*big, action)
    moved = 0
    for i in range(0, n, piece):
        part = [t[i:i + piece].clone() for t in (black, white, player, status)] + [torch.zeros(min(piece, n - i), dtype=torch.int64, device=DEV)]
        bb.step_batch(*part, action[i:i + piece].clone())
        for name, a, b in zip(("black", "white", "player", "status", "legal"), big, part):
            assert torch.equal(a[i:i + piece], b), (name, i)
        moved += int((part[0] != black[i:i + piece]).sum())
    assert moved > n // 8                                  # (the batch did step: moves that flip something are common on these boards)


@pytest.mark.parametrize("env", [{}, {"RAZ_SWEEP_HYBRID_WAVES": "100"}, {"RAZ_SWEEP_SLICED_STEP": "1"}],
                         ids=["as_adopted_for_large_batches", "hybrid_step_by_100_waves", "everything_sliced_step"])
def test_bit_sliced_forms_of_the_sweep_kernels_in_a_process_of_their_own(env):
    """The library runs the whole superblocks (2048 boards) of LARGE batches on the bit-sliced kernels of csrc/raz_sweep_sliced.h -
    k_legal_moves_sliced from 2^25 boards on, k_step_hybrid from 2^26 - and reads the thresholds once per process: every test of this
    file again in a child with RAZ_SWEEP_SLICED_MIN=2048 (every whole superblock sliced), the step kernel (a) as adopted, (b) by 100
    waves so that each walks many superblocks, (c) in the everything-sliced form kept beside it (RAZ_SWEEP_SLICED_STEP=1)."""
    if os.environ.get("RAZ_SWEEP_TEST_CHILD"):
        pytest.skip("a child of this test")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "not bit_sliced_forms"],
                       env={**{k: v for k, v in os.environ.items() if not k.startswith("RAZ_SWEEP_")}, "RAZ_SWEEP_SLICED_MIN": "2048", "RAZ_SWEEP_TEST_CHILD": "1", **env},
                       capture_output=True, text=True, timeout=1500,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]

