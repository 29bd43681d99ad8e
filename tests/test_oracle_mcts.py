"""CPU: the C MCTS/self-play oracle (oracle/orc_mcts.c + orc_net.c + orc_rng.c) against full games
played by the UNMODIFIED reference (tests/golden/mcts_games.json) — bit-exact: every action, every
root N and W (float64), ActionWithEvaluation n/q, every saved policy, resignation flags, and the
play_*.json rows the reference worker wrote (sha256 of the JSON text)."""
import ctypes
import hashlib
import json

import numpy as np
import pytest

import oracle as O
from oracle_util import load_mcts_golden, load_par_golden, orc_cfg_of, golden_net_blob, config_of, dense, rows_of_game


@pytest.fixture(scope="module")
def golden():
    return load_mcts_golden()


@pytest.fixture(scope="module")
def blob(golden):
    return golden_net_blob(golden["net"])


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    import ctypes
    lib = O.load_ext()
    U4, U2 = ctypes.c_uint32 * 4, ctypes.c_uint32 * 2
    kats = [([0] * 4, [0] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
            ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
            ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
             [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for c, k, exp in kats:
        out = U4()
        lib.orc_philox4x32_10(U4(*c), U2(*k), out)
        assert list(out) == exp


def test_det_math_accuracy():
    import math
    import numpy as np
    lib = O.load_ext()
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.random(3000), 10 ** rng.uniform(-300, 300, 3000)]):
        assert abs(lib.orc_det_log(x) - math.log(x)) <= 1e-15 * max(1.0, abs(math.log(x)))
    for x in rng.uniform(-700, 700, 5000):
        assert abs(lib.orc_det_exp(x) - math.exp(x)) <= 1e-15 * math.exp(x)
    for x in rng.uniform(-80, 20, 5000).astype(np.float32):
        assert abs(lib.orc_det_expf(float(x)) - math.exp(float(x))) <= 3e-7 * math.exp(float(x))
    for x in rng.uniform(-12, 12, 5000).astype(np.float32):
        assert abs(lib.orc_det_tanhf(float(x)) - math.tanh(float(x))) <= 3e-7


def test_dirichlet_noise_of_mask_properties():
    """The reference's own property test (test/lib/test_bitboard.py:115-122) on the injected sampler."""
    import ctypes
    lib = O.load_ext()
    mask, out = 47289423, (ctypes.c_double * 64)()
    for ev in range(50):
        lib.orc_dirichlet_noise_of_mask(mask, 0.5, 1, 2, ev, ctypes.byref(out))
        assert abs(sum(out) - 1.0) < 1e-12
        for i in range(64):
            assert (out[i] > 0) == bool(mask >> i & 1)


def test_numpy_pairwise_sum_restatement():
    """select_action's np.sum(float32[64]) is restated as numpy's 8-lane pairwise order."""
    import numpy as np
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = (rng.random(64) * (rng.random(64) < 0.3)).astype(np.float32)
        r = [a[j] for j in range(8)]
        for i in range(8, 64, 8):
            for j in range(8):
                r[j] = np.float32(r[j] + a[i + j])
        s = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3])) +
                       np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
        assert s == np.sum(a)


def test_games_bit_exact_vs_reference(golden, blob):
    assert len(golden["games"]) >= 10
    check_games_bit_exact(golden, blob)


def test_dirichlet_alpha_games_bit_exact_vs_reference(blob):
    """dirichlet_alpha 0.3, 1.0 and 0.03 (lib/bitboard.py:162-171 with alphas other than the shipped 0.5: the general
    Gamma(alpha <= 1) sampler of raz-rng-v1, not the Box-Muller pairs): games of the unmodified reference == oracle."""
    alpha = load_mcts_golden("mcts_alpha_games.json")
    assert alpha["net"] == load_mcts_golden()["net"]
    assert {g["resolved_play"]["dirichlet_alpha"] for g in alpha["games"]} == {0.3, 1.0, 0.03}
    check_games_bit_exact(alpha, blob)
    check_play_rows(alpha, blob)


def test_parallel_search_games_bit_exact_vs_reference_on_virtual_time_loop(blob):
    """parallel_search_num = 2..16 (raz-sched-v1): the oracle's round schedule against the unmodified
    reference player run on the exact-virtual-time event loop - same bit-exact bar as above."""
    par = load_par_golden()
    assert par["net"] == load_mcts_golden()["net"]
    ks = {g["resolved_play"]["parallel_search_num"] for g in par["games"]}
    assert ks >= {2, 3, 4, 8, 16}
    check_games_bit_exact(par, blob)
    check_play_rows(par, blob)


def test_tree_carried_across_games_bit_exact_vs_reference(blob):
    """reset_mtcs_info_per_game = 3 (config/mini.yml as shipped, incl. parallel_search_num 4, solver, re-thinking):
    three consecutive games of one worker on ONE MCTSInfo (worker/self_play.py:109-111,132-134; each game's new
    players start with expanded = set(var_p.keys()), agent/player.py:47) == the reference's three games."""
    ser = load_mcts_golden("mcts_series_games.json")
    assert {g["resolved_play"]["reset_mtcs_info_per_game"] for g in ser["games"]} == {3}
    assert {g["resolved_play"]["parallel_search_num"] for g in ser["games"]} == {1, 4}
    check_games_bit_exact(ser, blob, series=True)


_played = {}   # (golden file, variant, game id) -> the oracle's game: played once, checked by several tests


def _oracle_game(golden, g, blob, tree=None):
    key = (golden["_generator"], golden.get("event_loop"), g["variant"], g["game_id"])
    if key not in _played:
        _played[key] = O.selfplay_game(orc_cfg_of(g), blob, g["seed"], g["game_id"], g["sims_per_move"], tree=tree)
    return _played[key]


def check_games_bit_exact(golden, blob, series=False):
    trees = {}
    for g in golden["games"]:
        tree = None
        if series:   # the games of a variant were played in order by one worker: one MCTSInfo
            assert g["series_index"] == (0 if g["variant"] not in trees else trees[g["variant"]][1] + 1)
            tree = trees[g["variant"]][0] if g["variant"] in trees else O.Tree()
            trees[g["variant"]] = (tree, g["series_index"])
        plies, summ = _oracle_game(golden, g, blob, tree)
        tag = f'{g["variant"]}/{g["game_id"]}'
        assert [p["action"] for p in plies] == [p["action"] for p in g["plies"]], tag
        assert summ["winner"] == g["winner"] and summ["turn"] == g["turn"], tag
        assert (summ["black"], summ["white"]) == (int(g["black"], 16), int(g["white"], 16)), tag
        assert (bool(summ["resigned_black"]), bool(summ["resigned_white"])) == \
            (g["resigned_black"], g["resigned_white"]), tag
        assert summ["n_expand"] == g["nn_positions"], tag
        for i, (a, b) in enumerate(zip(plies, g["plies"])):
            assert a["player"] == b["player"] and a["own"] == int(b["own"], 16) and a["enemy"] == int(b["enemy"], 16)
            assert a["root_n"] == dense(b["root_n"]), (tag, i)
            assert a["root_w"] == dense(b["root_w"]), (tag, i)
            assert a["has_row"] == b["has_row"], (tag, i)
            assert a["solved"] == b.get("solved", False), (tag, i)
            if a["action"] >= 0:
                assert a["n"] == b["n"] and a["q"] == b["q"], (tag, i)
            if b["has_row"]:
                assert a["saved_policy"] == dense(b["saved_policy"]), (tag, i)
        if not config_of(g).play.share_mtcs_info_in_self_play:
            assert summ["n_mirror_hits"] == 0  # colour-swapped transpositions never occurred


def test_play_rows_identical_to_reference_files(golden, blob):
    check_play_rows(golden, blob)


def check_play_rows(golden, blob):
    for g in golden["games"]:
        plies, summ = _oracle_game(golden, g, blob)
        rows = rows_of_game(plies, summ["winner"])
        dropped = summ["winner"] == 3 and not (g["resolved_play_data"]["drop_draw_game_rate"] <= summ["drop_draw_u"])
        if g["play_rows_sha256"] is None:
            assert dropped or not rows
            continue
        assert len(rows) == g["play_rows_count"]
        assert rows[:9] == [[list(r[0]), r[1], r[2]] for r in g["play_rows_head"]]
        assert hashlib.sha256(json.dumps(rows).encode()).hexdigest() == g["play_rows_sha256"], g["variant"]


@pytest.mark.needs_reference
def test_live_reference_game_matches_oracle(blob):
    """A fresh differential run (new seed/config) against the imported reference, container only."""
    import ref_harness as rh
    import ref_selfplay as rs
    cfg = rh.load_config("alpha_go_zero.yml", {"play": {"parallel_search_num": 1, "c_puct": 1.5, "noise_eps": 0.4,
                                                         "dirichlet_alpha": 1.0, "change_tau_turn": 6}})
    ref = rs.run_reference_game(cfg, blob, seed=99, game_id=123456, sims_per_move=15)
    plies, summ = O.selfplay_game(O.play_cfg_from_config(cfg), blob, 99, 123456, 15)
    assert [p["action"] for p in plies] == [p["action"] for p in ref["plies"]]
    for a, b in zip(plies, ref["plies"]):
        assert a["root_n"] == b["root_n"] and a["root_w"] == b["root_w"]
    assert summ["winner"] == ref["winner"]


def test_gamma_sampler_above_one_has_the_gamma_distribution():
    """raz-rng-v1's Gamma(alpha > 1) (numpy's legacy Marsaglia-Tsang scheme on counter-based draws, oracle/orc_rng.c): a
    Kolmogorov-Smirnov test against scipy's Gamma(alpha) over 20 000 samples per alpha, and Dirichlet components that sum to 1."""
    from scipy import stats
    lib = O.load_ext()
    for alpha in (1.01, 1.7, 3.0, 12.5):
        xs = np.array([lib.orc_gamma_sample(alpha, 11, 5, ev, ev % 7) for ev in range(20000)])
        assert xs.min() > 0.0
        assert stats.kstest(xs, "gamma", args=(alpha,)).pvalue > 1e-3, alpha
    out = (ctypes.c_double * 64)()
    lib.orc_dirichlet_noise_of_mask(0x0000001818000000 | 0x81, 2.5, 1, 2, 3, ctypes.byref(out))
    v = np.array(out[:])
    assert abs(v.sum() - 1.0) < 1e-12 and (v > 0).sum() == 6


@pytest.mark.needs_reference
def test_reference_game_at_dirichlet_alpha_above_one_matches_oracle(blob):
    """The unmodified reference with dirichlet_alpha = 2.0 (np.random.dirichlet served by the raz-rng-v1 stream) == the oracle."""
    import ref_harness as rh
    import ref_selfplay as rs
    cfg = rh.load_config("alpha_go_zero.yml", {"play": {"parallel_search_num": 1, "noise_eps": 0.5, "dirichlet_alpha": 2.0}})
    ref = rs.run_reference_game(cfg, blob, seed=31, game_id=77, sims_per_move=12)
    plies, summ = O.selfplay_game(O.play_cfg_from_config(cfg), blob, 31, 77, 12)
    assert [p["action"] for p in plies] == [p["action"] for p in ref["plies"]]
    for a, b in zip(plies, ref["plies"]):
        assert a["root_n"] == b["root_n"] and a["root_w"] == b["root_w"]


@pytest.mark.needs_reference
def test_game_taken_up_at_a_position_matches_the_reference(blob):
    """orc_selfplay_game_from (bench.py's check of its steady-state batch): the reference worker whose env is put on a
    mid-game position by ReversiEnv.update (env/reversi_env.py:33-40) instead of reset() plays, with fresh players and
    fresh random streams, the game the oracle plays from that position - white and black to move, with a pass on the way."""
    import ref_harness as rh
    import ref_selfplay as rs
    cfg = rh.load_config("ch5.yml", {"play": {"parallel_search_num": 1, "thinking_loop": 1, "use_solver_turn": 0,
                                              "use_solver_turn_in_simulation": 0}})
    env = O.OrcEnv()
    lib = O.load()
    rng = np.random.default_rng(3)
    starts = []
    for plies_in in (7, 30, 52):   # reach positions by random playouts
        lib.orc_env_reset(ctypes.byref(env))
        for _ in range(plies_in):
            own, enemy = (env.black, env.white) if env.next_player == 1 else (env.white, env.black)
            legal = lib.orc_find_correct_moves(own, enemy)
            moves = [i for i in range(64) if legal >> i & 1]
            lib.orc_env_step(ctypes.byref(env), int(rng.choice(moves)))
            assert not env.done
        starts.append((int(env.black), int(env.white), int(env.next_player)))
    assert {s[2] for s in starts} == {1, 2}
    for k, st in enumerate(starts):
        ref = rs.run_reference_game(cfg, blob, seed=5, game_id=900 + k, sims_per_move=14, start=st)
        plies, summ = O.selfplay_game(O.play_cfg_from_config(cfg), blob, 5, 900 + k, 14, start=st)
        assert [p["action"] for p in plies] == [p["action"] for p in ref["plies"]] and len(plies) >= 5
        for a, b in zip(plies, ref["plies"]):
            assert a["root_n"] == b["root_n"] and a["root_w"] == b["root_w"] and a["player"] == b["player"]
        assert summ["winner"] == ref["winner"]
    # from the initial position it is the ordinary game
    a = O.selfplay_game(O.play_cfg_from_config(cfg), blob, 5, 77, 9, start=(0x0000000810000000, 0x0000001008000000, 1))
    assert a == O.selfplay_game(O.play_cfg_from_config(cfg), blob, 5, 77, 9)


@pytest.mark.needs_reference
def test_live_reference_parallel_search_matches_oracle(blob):
    """Fresh differential runs at parallel_search_num 5 and 7 (not in the goldens), container only."""
    import ref_harness as rh
    import ref_selfplay as rs
    for k, yml, seed in ((5, "mini.yml", 77), (7, "alpha_go_zero.yml", 78)):
        cfg = rh.load_config(yml, {"play": {"parallel_search_num": k, "reset_mtcs_info_per_game": 1,
                                            "use_solver_turn": 0, "use_solver_turn_in_simulation": 0}})
        ref = rs.run_reference_game(cfg, blob, seed=seed, game_id=4242, sims_per_move=18, virtual_time=True)
        plies, summ = O.selfplay_game(O.play_cfg_from_config(cfg, parallel_search_num=k), blob, seed, 4242, 18)
        assert [p["action"] for p in plies] == [p["action"] for p in ref["plies"]]
        for a, b in zip(plies, ref["plies"]):
            assert a["root_n"] == b["root_n"] and a["root_w"] == b["root_w"]
        assert summ["n_expand"] == ref["nn_positions"] and summ["winner"] == ref["winner"]


@pytest.mark.needs_reference
def test_virtual_time_schedule_is_representative_of_the_real_event_loop():
    """raz-sched-v1 is the reference's event loop with computation taking no time; on the REAL loop the
    interleaving depends on wall-clock timers and differs run to run, so only distributions are comparable
    (tests/golden/par_vs_realtime.py): NN leaves per simulation and root concentration agree within 20 %."""
    import importlib.util
    import os
    import ref_harness as rh
    import ref_selfplay as rs
    spec = importlib.util.spec_from_file_location("par_vs_realtime", os.path.join(os.path.dirname(__file__), "golden", "par_vs_realtime.py"))
    pvr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pvr)
    blob = golden_net_blob(load_mcts_golden()["net"])
    over = {"play": {"parallel_search_num": 4, "use_solver_turn": 0, "use_solver_turn_in_simulation": 0,
                     "reset_mtcs_info_per_game": 1, "thinking_loop": 1}}
    st = {}
    for loop, vt in (("real", False), ("virtual", True)):
        games = [rs.run_reference_game(rh.load_config("mini.yml", over), blob, 91, gid, 24, virtual_time=vt) for gid in range(3)]
        st[loop] = pvr.stats(games, 24)
    for k in ("leaves/sim", "top share", "entropy"):
        assert abs(st["real"][k] - st["virtual"][k]) <= 0.2 * abs(st["virtual"][k]), (k, st)
