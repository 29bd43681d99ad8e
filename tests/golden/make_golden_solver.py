#!/usr/bin/env python
"""Generate tests/golden/solver_kat.json: answers of the reference's COMPILED Cython end-game solver
(lib/alt/reversi_solver_cython.pyx ReversiSolver.solve, pyximport) - build container only:
    python tests/golden/make_golden_solver.py
Contents: the reference's own three known answers (lib/reversi_solver.py:102-156: q1 -> (57, +2) non-exact,
q2 -> (4 or 14, -2), q3 -> (3, +2) exact) and 120 late-game positions (<= 10 empties, reached by seeded random
playouts of the reference env) solved in both modes by a FRESH solver each (the reference's memo never changes
an answer, only its cost).  Every number is produced by reference code."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402

Q1 = """
##########
#XXXX    #
#XOXX    #
#XOXXOOOO#
#XOXOXOOO#
#XOXXOXOO#
#OOOOXOXO#
# OOOOOOO#
#  XXXXXO#
##########"""
Q2 = """
##########
#XXXX    #
#XXXX X  #
#XXXXXXOO#
#XXXXXXOO#
#XXXXOXOO#
#OXOOXOXO#
# OOOOOOO#
#OOOOOOOO#
##########"""
Q3 = """
##########
#  X OOO #
#X XOXO O#
#XXXXOXOO#
#XOXOOXXO#
#XOOOOXXO#
#XOOOXXXO#
# OOOOXX #
#  OOOOX #
##########"""


def main():
    rh.install()
    import pyximport
    pyximport.install(build_dir="/tmp/pyxbld_raz", language_level=3)
    from reversi_zero.lib.alt.reversi_solver_cython import ReversiSolver
    from reversi_zero.lib.util import parse_to_bitboards
    from reversi_zero.lib.bitboard import find_correct_moves, bit_count
    from reversi_zero.env.reversi_env import ReversiEnv, Player

    def solve(b, w, player, exactly):
        m, s = ReversiSolver().solve(b, w, Player(player), timeout=300, exactly=exactly)
        return [m, s]

    out = {"_generator": "tests/golden/make_golden_solver.py", "kat": [], "positions": []}
    for name, board, player, exactly, expect in (("q1", Q1, 2, False, "(57, +2)"), ("q2", Q2, 1, False, "(4 or 14, -2)"),
                                                 ("q3", Q3, 2, True, "(3, +2)")):
        b, w = parse_to_bitboards(board)
        out["kat"].append({"name": name, "black": "0x%016x" % b, "white": "0x%016x" % w, "next_player": player,
                           "exactly": exactly, "reference_comment": expect, "answer": solve(b, w, player, exactly),
                           "answer_other_mode": solve(b, w, player, not exactly)})
        print(name, out["kat"][-1]["answer"], out["kat"][-1]["answer_other_mode"], "expected", expect)
    rng = random.Random(20240)
    while len(out["positions"]) < 120:
        env = ReversiEnv().reset()
        stop_at = 64 - rng.randint(3, 10)   # discs on the board when we stop: 4..12 empties
        while not env.done and bit_count(env.board.black) + bit_count(env.board.white) < stop_at:
            own, enemy = env.get_own_and_enemy()
            legal = find_correct_moves(own, enemy)
            env.step(rng.choice([i for i in range(64) if legal >> i & 1]))
        if env.done:
            continue
        b, w, pl = env.board.black, env.board.white, env.next_player.value
        out["positions"].append({"black": "0x%016x" % b, "white": "0x%016x" % w, "next_player": pl,
                                 "empties": 64 - bit_count(b) - bit_count(w),
                                 "exact": solve(b, w, pl, True), "non_exact": solve(b, w, pl, False)})
    path = os.path.join(HERE, "solver_kat.json")
    with open(path, "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, len(out["positions"]), "positions")


if __name__ == "__main__":
    main()
