#!/usr/bin/env python
"""Generate tests/golden/bitboard_env.json from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
Everything written comes out of reference functions (lib/bitboard.py, env/reversi_env.py,
lib/util.py, agent/player.py add_data_to_move_buffer_with_8_symmetries) — this script contains no
game logic of its own beyond choosing inputs.  The ASCII boards are the ones the reference's own
tests use (test/lib/test_bitboard.py:11-112), read from that file at generation time.
"""
import json
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402

rh.install()
import numpy as np  # noqa: E402
from reversi_zero.lib import bitboard as rb  # noqa: E402
from reversi_zero.lib.util import parse_to_bitboards  # noqa: E402
from reversi_zero.env.reversi_env import ReversiEnv, Player, Winner  # noqa: E402


def hx(x):
    return "0x%016x" % x


def legal_list(m):
    return [i for i in range(64) if m >> i & 1]


def reference_test_boards():
    """The `ex` boards of test/lib/test_bitboard.py (parsed out of the file, not retyped)."""
    src = open(os.path.join(rh.REFERENCE_ROOT, "test", "lib", "test_bitboard.py")).read()
    boards = re.findall(r"ex = '''\n(##########\n(?:#.{8}#\n){8}##########)'''", src)
    assert len(boards) == 3, len(boards)
    return boards


def position_record(own, enemy):
    m = rb.find_correct_moves(own, enemy)
    return {"own": hx(own), "enemy": hx(enemy), "legal": hx(m),
            "flips": {str(a): hx(rb.calc_flip(a, own, enemy)) for a in legal_list(m)}}


def main():
    rng = random.Random(20240924)
    out = {"_generator": "tests/golden/make_golden.py", "_source": "unmodified reference, lib/bitboard.py + env/reversi_env.py"}

    # 1. the reference's own test boards, both colours to move
    tb = []
    for s in reference_test_boards():
        b, w = parse_to_bitboards(s)
        tb.append({"ascii": s, "black": hx(b), "white": hx(w),
                   "black_to_move": position_record(b, w), "white_to_move": position_record(w, b)})
    env = ReversiEnv().reset()
    tb.append({"ascii": "initial", "black": hx(env.board.black), "white": hx(env.board.white),
               "black_to_move": position_record(env.board.black, env.board.white),
               "white_to_move": position_record(env.board.white, env.board.black)})
    out["test_boards"] = tb

    # 2. random playouts: random.choice over the ascending legal list
    playouts, harvested = [], []
    for g in range(40):
        env = ReversiEnv().reset()
        actions, players = [], []
        while not env.done:
            own, enemy = env.get_own_and_enemy()
            moves = legal_list(rb.find_correct_moves(own, enemy))
            harvested.append((own, enemy))
            a = rng.choice(moves)
            actions.append(a)
            players.append(env.next_player.value)
            env.step(a)
        playouts.append({"actions": actions, "players": players, "black": hx(env.board.black),
                         "white": hx(env.board.white), "winner": env.winner.value, "turn": env.turn})
    out["playouts"] = playouts

    # 3. positions harvested from those games at random plies, every legal flip
    rng.shuffle(harvested)
    out["positions"] = [position_record(o, e) for o, e in harvested[:600]]

    # 4. arbitrary 64-bit garbage (overlapping colours, occupied squares): total-function parity
    garbage = []
    for _ in range(600):
        own, enemy, pos = rng.getrandbits(64), rng.getrandbits(64), rng.randrange(64)
        if rng.random() < 0.5:
            enemy &= ~own
        garbage.append({"own": hx(own), "enemy": hx(enemy), "pos": pos,
                        "legal": hx(rb.find_correct_moves(own, enemy)),
                        "flip": hx(rb.calc_flip(pos, own, enemy))})
    out["garbage"] = garbage

    # 5. symmetries and small helpers
    syms = []
    for _ in range(200):
        x = rng.getrandbits(64)
        syms.append({"x": hx(x), "flip_vertical": hx(rb.flip_vertical(x)),
                     "flip_diag_a1h8": hx(rb.flip_diag_a1h8(x)), "rotate90": hx(rb.rotate90(x)),
                     "rotate180": hx(rb.rotate180(x)), "bit_count": rb.bit_count(x),
                     "bit_to_array": "".join(str(v) for v in rb.bit_to_array(x, 64))})
    out["symmetries"] = syms

    # 6. env edge cases: resign, illegal move, update()/Board() zero quirk, pass, early end
    edge = []
    for desc, (b, w, p), act in [
        ("resign_black", (None, None, Player.black), None),
        ("resign_white", (None, None, Player.white), None),
        ("illegal_black", (None, None, Player.black), 0),
        ("illegal_white", (None, None, Player.white), 63),
        ("occupied_square", (None, None, Player.black), 27),
    ]:
        env = ReversiEnv().reset()
        env.next_player = p
        env.step(act)
        edge.append({"desc": desc, "black_in": hx(0x0000000810000000), "white_in": hx(0x0000001008000000),
                     "player_in": p.value, "action": -1 if act is None else act,
                     "black": hx(env.board.black), "white": hx(env.board.white),
                     "next_player": env.next_player.value, "turn": env.turn, "done": env.done,
                     "winner": env.winner.value if env.winner else 0})
    env = ReversiEnv().update(0, 0, Player.white)
    edge.append({"desc": "update_zero_boards", "black": hx(env.board.black), "white": hx(env.board.white),
                 "turn": env.turn})
    out["env_edge"] = edge

    # 7. 8-symmetry training rows (agent/player.py:166-179), incl. the reference test's own case
    class _API:
        def predict(self, x):
            raise AssertionError("not used")
    cfg = rh.load_config()
    rows = []
    cases = [((1 << 0) | (1 << 9), (1 << 55) | (1 << 63), {7: 0.8, 56: 0.2})]  # test_player.py:30-35
    for _ in range(12):
        own = rng.getrandbits(64)
        enemy = rng.getrandbits(64) & ~own
        pol = {rng.randrange(64): rng.random() for _ in range(5)}
        cases.append((own, enemy, pol))
    for own, enemy, pol in cases:
        player = rh.make_player(cfg, _API())
        policy = np.zeros(64)
        for k, v in pol.items():
            policy[k] = v
        player.add_data_to_move_buffer_with_8_symmetries(own, enemy, policy)
        rows.append({"own": hx(own), "enemy": hx(enemy), "policy": {str(k): v for k, v in pol.items()},
                     "rows": [[hx(o), hx(e), [float(x) for x in p]] for (o, e), p in player.moves]})
    out["sym8_rows"] = rows

    path = os.path.join(HERE, "bitboard_env.json")
    with open(path, "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
