"""The end-game solver (lib/alt/reversi_solver_cython.pyx; agent/player.py:100-103,150-161,237-251) pinned directly:
tests/golden/solver_kat.json holds the answers of the reference's COMPILED Cython solver (make_golden_solver.py) for the
reference's own three known answers (lib/reversi_solver.py:102-156: q1 -> (57, +2), q2 -> (4 or 14, -2), q3 -> (3, +2))
and for 120 late-game positions in both modes.  CPU: oracle/orc_solver.c == those answers; GPU: the device solver
(solver_solve in csrc/raz_engine.hip, reached through raz_engine_set_position on the KAT boards) == those answers;
needs_reference: a fresh sweep against the live Cython solver."""
import ctypes
import json
import os
import types

import numpy as np
import pytest

import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solver_kat.json")


@pytest.fixture(scope="module")
def kat():
    with open(GOLDEN) as f:
        return json.load(f)


def _orc_solve(black, white, player, exactly):
    lib = O.load_ext()
    s = lib.orc_solver_new()
    mv, sc = ctypes.c_int(-1), ctypes.c_int(0)
    ok = lib.orc_solver_solve(s, black, white, player, int(exactly), ctypes.byref(mv), ctypes.byref(sc))
    lib.orc_solver_free(s)
    return [mv.value, sc.value] if ok else [None, None]


def test_reference_known_answers(kat):
    """q1/q2/q3 of lib/reversi_solver.py:102-156, as the reference's comments state them and as its compiled solver answers."""
    want = {"q1": ({57}, 2), "q2": ({4, 14}, -2), "q3": ({3}, 2)}
    for k in kat["kat"]:
        moves, score = want[k["name"]]
        assert k["answer"][0] in moves and k["answer"][1] == score, k          # the golden itself says what the reference prints
        got = _orc_solve(int(k["black"], 16), int(k["white"], 16), k["next_player"], k["exactly"])
        assert got == k["answer"], (k["name"], got)
        assert _orc_solve(int(k["black"], 16), int(k["white"], 16), k["next_player"], not k["exactly"]) == k["answer_other_mode"]


def test_oracle_solver_equals_compiled_cython_on_golden_positions(kat):
    assert len(kat["positions"]) >= 100
    for p in kat["positions"]:
        b, w, pl = int(p["black"], 16), int(p["white"], 16), p["next_player"]
        assert _orc_solve(b, w, pl, True) == p["exact"], p
        assert _orc_solve(b, w, pl, False) == p["non_exact"], p


@pytest.mark.needs_reference
def test_oracle_solver_vs_live_cython_sweep():
    """Fresh positions (not in the golden file) against the imported, compiled reference solver - container only."""
    import random
    import ref_harness as rh
    rh.install()
    import pyximport
    pyximport.install(build_dir="/tmp/pyxbld_raz", language_level=3)
    from reversi_zero.lib.alt.reversi_solver_cython import ReversiSolver
    from reversi_zero.env.reversi_env import ReversiEnv, Player
    from reversi_zero.lib.bitboard import find_correct_moves, bit_count
    rng = random.Random(777)
    done = 0
    while done < 40:
        env = ReversiEnv().reset()
        stop_at = 64 - rng.randint(2, 9)
        while not env.done and bit_count(env.board.black) + bit_count(env.board.white) < stop_at:
            own, enemy = env.get_own_and_enemy()
            legal = find_correct_moves(own, enemy)
            env.step(rng.choice([i for i in range(64) if legal >> i & 1]))
        if env.done:
            continue
        b, w, pl = env.board.black, env.board.white, env.next_player.value
        for exactly in (True, False):
            m, s = ReversiSolver().solve(b, w, Player(pl), timeout=300, exactly=exactly)
            assert _orc_solve(b, w, pl, exactly) == [m, s]
        done += 1


@pytest.mark.gpu
def test_device_solver_equals_compiled_cython(kat):
    """The device solver at the root (exact mode, agent/player.py:100-103,150-161): every golden position and the three
    KAT boards armed as one move on an engine slot; the move must be the Cython solver's and q = sign(score), n = 999."""
    import torch  # noqa: F401
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    play = types.SimpleNamespace(
        simulation_num_per_move=8, share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=40,
        start_rethinking_turn=10, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=10, virtual_loss=3,
        parallel_search_num=1, resign_threshold=None, allowed_resign_turn=10, disable_resignation_rate=0.0,
        use_solver_turn=46, use_solver_turn_in_simulation=46)
    cfg = types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
    cases = [(int(k["black"], 16), int(k["white"], 16), k["next_player"], k["answer"] if k["exactly"] else k["answer_other_mode"])
             for k in kat["kat"]]
    cases += [(int(p["black"], 16), int(p["white"], 16), p["next_player"], p["exact"]) for p in kat["positions"]]
    cases = [c for c in cases if bin(c[0] | c[1]).count("1") - 4 >= 46]   # the engine's declared limit: <= 14 empties
    n = len(cases)
    assert n >= 100
    eng = SelfPlayEngine(cfg, DeviceNet(ReversiNet(16, 1, 16).keras_init_(0).to_blob(), "cuda:0"), n_games=n, seed=3, sims_hint=8)
    eng.start(0, 8)
    for g, (b, w, pl, _) in enumerate(cases):
        eng.set_position(g, b, w, pl, 8, enable_resign=False, one_move=True)
    for _ in range(4000):      # a 10-empties exact solve takes many launches: it runs on a per-launch budget and is parked in between
        eng.step(4)
        if eng.stats()["idle_or_done"] >= len(cases):
            break
    raw = eng.read_raw()
    for g, (b, w, pl, (move, score)) in enumerate(cases):
        assert int(raw["n_plies"][g]) == 1, g
        h = raw["headers"][g, 0]
        assert int(h["flags"]) & 1, (g, "not solved")
        assert int(h["action"]) == move, (g, int(h["action"]), move)
        assert float(h["n"]) == 999.0 and float(h["q"]) == float(np.sign(score)), (g, float(h["q"]), score)
