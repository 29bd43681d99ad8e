"""CPU: the C oracle (oracle/orc_bitboard.c) against the committed golden vectors, which were
produced by the unmodified reference (tests/golden/make_golden.py), and against the hex constants
SURVEY.md §8(c) lists for the reference's own test boards."""
import ctypes

import pytest

from conftest import H
import oracle as O


def test_initial_position_constants(orc):
    e = O.OrcEnv()
    orc.orc_env_reset(ctypes.byref(e))
    assert e.black == 0x0000000810000000 and e.white == 0x0000001008000000
    assert orc.orc_find_correct_moves(e.black, e.white) == 0x0000102004080000  # {19,26,37,44}
    assert orc.orc_calc_flip(19, e.black, e.white) == 0x0000000008000000


def test_reference_test_boards(orc, golden_bb):
    survey = {  # SURVEY.md §8(c), computed from test/lib/test_bitboard.py
        0: (0x00000000081d0603, 0x0002043814020100, 0x0000780623000000, None),
        1: (0x0088ffabd5dfdf5f, 0x000700542a202020, 0x0f00000000000000, 0x0870000000000080),
        2: (0xfe71797106000203, 0x008e868ef9fffd7c, 0x0000000000000080, 0x0100000000000000),
    }
    for i, tb in enumerate(golden_bb["test_boards"]):
        b, w = H(tb["black"]), H(tb["white"])
        if i in survey:
            sb, sw, mb, mw = survey[i]
            assert (b, w) == (sb, sw)
            assert orc.orc_find_correct_moves(b, w) == mb
            if mw is not None:
                assert orc.orc_find_correct_moves(w, b) == mw
        for side, (own, enemy) in (("black_to_move", (b, w)), ("white_to_move", (w, b))):
            rec = tb[side]
            assert orc.orc_find_correct_moves(own, enemy) == H(rec["legal"])
            for a, f in rec["flips"].items():
                assert orc.orc_calc_flip(int(a), own, enemy) == H(f)
    assert orc.orc_calc_flip(7, 0xfe71797106000203, 0x008e868ef9fffd7c) == 0x008080808080807c


def test_positions_and_garbage(orc, golden_bb):
    for rec in golden_bb["positions"]:
        own, enemy = H(rec["own"]), H(rec["enemy"])
        assert orc.orc_find_correct_moves(own, enemy) == H(rec["legal"])
        for a, f in rec["flips"].items():
            assert orc.orc_calc_flip(int(a), own, enemy) == H(f)
    for rec in golden_bb["garbage"]:
        own, enemy = H(rec["own"]), H(rec["enemy"])
        assert orc.orc_find_correct_moves(own, enemy) == H(rec["legal"])
        assert orc.orc_calc_flip(rec["pos"], own, enemy) == H(rec["flip"])


def test_symmetries(orc, golden_bb):
    buf = (ctypes.c_uint8 * 64)()
    for rec in golden_bb["symmetries"]:
        x = H(rec["x"])
        for name in ("flip_vertical", "flip_diag_a1h8", "rotate90", "rotate180"):
            assert getattr(orc, "orc_" + name)(x) == H(rec[name]), name
        assert orc.orc_bit_count(x) == rec["bit_count"]
        orc.orc_bit_to_array(x, 64, buf)
        assert "".join(str(v) for v in buf) == rec["bit_to_array"]


def test_playouts(orc, golden_bb):
    for g in golden_bb["playouts"]:
        e = O.OrcEnv()
        orc.orc_env_reset(ctypes.byref(e))
        for a, p in zip(g["actions"], g["players"]):
            assert not e.done and e.next_player == p
            orc.orc_env_step(ctypes.byref(e), a)
        assert e.done and e.winner == g["winner"] and e.turn == g["turn"]
        assert (e.black, e.white) == (H(g["black"]), H(g["white"]))


def test_env_edges(orc, golden_bb):
    for rec in golden_bb["env_edge"]:
        e = O.OrcEnv()
        if rec["desc"] == "update_zero_boards":
            orc.orc_env_update(ctypes.byref(e), 0, 0, 2)
            assert (e.black, e.white, e.turn) == (H(rec["black"]), H(rec["white"]), rec["turn"])
            continue
        orc.orc_env_reset(ctypes.byref(e))
        e.next_player = rec["player_in"]
        orc.orc_env_step(ctypes.byref(e), rec["action"])
        assert (e.black, e.white) == (H(rec["black"]), H(rec["white"]))
        assert (e.next_player, e.turn, bool(e.done), e.winner) == \
            (rec["next_player"], rec["turn"], rec["done"], rec["winner"])


@pytest.mark.needs_reference
def test_oracle_vs_live_reference_random(orc):
    """Differential run against the imported reference on fresh random inputs (container only)."""
    import random
    import ref_harness as rh
    rh.install()
    from reversi_zero.lib import bitboard as rb
    rng = random.Random(7)
    for _ in range(3000):
        own, enemy, pos = rng.getrandbits(64), rng.getrandbits(64), rng.randrange(64)
        if rng.random() < 0.7:
            enemy &= ~own
        assert orc.orc_find_correct_moves(own, enemy) == rb.find_correct_moves(own, enemy)
        assert orc.orc_calc_flip(pos, own, enemy) == rb.calc_flip(pos, own, enemy)
