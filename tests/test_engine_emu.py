"""CPU: the tree kernels of csrc/raz_engine.hip (k_tree, k_tree_par, k_gc, the harvest / adopt / read-node kernels) stepped by
the WAVE EMULATOR (tests/native/wave_emu: workgroups as cooperative fibers, the 64 lanes of a wave meeting at every readlane /
DPP / ballot) against the CPU oracle, bit for bit, at sizes a CPU finishes in seconds.  Leaves are evaluated by the oracle's C
net (bit-identical to the exact-f32 device kernels).  This is the build container's debugging loop for the kernels' LOGIC -
node layout, pool / table / pruning, the simulation-slot schedule; the parity tests of record are the GPU tests through the
real libraz.so (tests/test_engine_gpu.py, test_engine_par_gpu.py, test_continuous_gpu.py).  The emulator also verifies what
the hardware would not: that every cross-lane operation is reached by all lanes of its wave together."""
import numpy as np
import pytest

import oracle as O
from emu_util import EmuEngine
from oracle_util import load_mcts_golden, load_par_golden, golden_net_blob, config_of, dense


@pytest.fixture(scope="module")
def golden():
    return load_mcts_golden()


@pytest.fixture(scope="module")
def blob(golden):
    return golden_net_blob(golden["net"])


def _same(tag, plies, summ, ref_plies, ref_winner):
    assert len(plies) == len(ref_plies), (tag, len(plies), len(ref_plies))
    for i, (a, b) in enumerate(zip(plies, ref_plies)):
        for k in ("player", "own", "enemy", "action", "has_row"):
            assert a[k] == b[k], (tag, i, k, a[k], b[k])
        assert a["root_n"] == b["root_n"], (tag, i, "root_n")
        assert a["root_w"] == b["root_w"], (tag, i, "root_w")
        assert a["solved"] == b.get("solved", False), (tag, i)
        if a["action"] >= 0:
            assert a["n"] == b["n"] and a["q"] == b["q"], (tag, i)
    assert summ["winner"] == ref_winner, tag


def _variant(golden, name):
    return next(g for g in golden["games"] if g["variant"] == name)


def test_emulated_kernel_replays_reference_golden_games(golden, blob):
    """Two of the unmodified reference's golden games (shared tree with re-thinking loops; unshared AlphaGoZero tree)."""
    done = set()
    for g in golden["games"]:
        if g["variant"] in done or g["variant"] not in ("mini_shared", "agz") or g["sims_per_move"] > 40:
            continue
        done.add(g["variant"])
        cfg = config_of(g)
        eng = EmuEngine(cfg, blob, n_games=1, seed=g["seed"], sims_hint=g["sims_per_move"])
        eng.start(g["game_id"], g["sims_per_move"])
        eng.run(chunk=64)
        (plies, summ), = eng.records(save_policy_of_tau_1=g["resolved_play_data"]["save_policy_of_tau_1"])
        ref = [dict(p, own=int(p["own"], 16), enemy=int(p["enemy"], 16), root_n=dense(p["root_n"]), root_w=dense(p["root_w"])) for p in g["plies"]]
        _same(g["variant"], plies, summ, ref, g["winner"])
        assert (summ["black"], summ["white"]) == (int(g["black"], 16), int(g["white"], 16))
    assert done == {"mini_shared", "agz"}


@pytest.mark.parametrize("variant,pool", [("mini_shared", None), ("mini_shared", 480), ("agz", 200)])
def test_emulated_batch_equals_oracle_with_and_without_pruning(golden, blob, variant, pool):
    """A small batch with mixed simulation counts == independent oracle games; with a node pool far too small for a whole
    game k_gc prunes several times per game and nothing changes."""
    cfg = config_of(_variant(golden, variant))
    n = 3
    sims = np.array([9, 12, 15], dtype=np.uint32)
    eng = EmuEngine(cfg, blob, n_games=n, seed=31, sims_hint=15, nodes_per_game=pool)
    eng.start(500, sims)
    eng.run(chunk=8 if pool else 32)
    assert pool is None or eng.gc_runs >= 2
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg)
    for i in range(n):
        plies, summ = O.selfplay_game(ocfg, blob, 31, 500 + i, int(sims[i]))
        _same(f"{variant}/{pool}/{i}", recs[i][0], recs[i][1], plies, summ["winner"])


@pytest.mark.parametrize("budget", [None, 5])
def test_emulated_solver_games_equal_oracle(golden, blob, budget, monkeypatch):
    """mini.yml as shipped: exact solver at the root, win/loss solver inside simulations (per-game memo).  budget: the solver
    iterations a game may spend per launch - with 5, every solve of more than a handful of nodes is suspended many times (the
    root's in begin_move, the ones inside simulations in the middle of a descent) and the records must not notice."""
    if budget is not None:
        monkeypatch.setenv("RAZ_SOLVER_BUDGET", str(budget))
    cfg = config_of(_variant(golden, "mini_solver_noresign"))
    eng = EmuEngine(cfg, blob, n_games=1, seed=41, sims_hint=10)
    eng.start(900, 10)
    eng.run(chunk=32)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg)
    solved = 0
    for i in range(1):
        plies, summ = O.selfplay_game(ocfg, blob, 41, 900 + i, 10)
        _same(f"solver/{i}", recs[i][0], recs[i][1], plies, summ["winner"])
        solved += sum(p["solved"] for p in plies)
    assert solved > 0


@pytest.mark.parametrize("k,pool", [(4, None), (3, 400)])
def test_emulated_slot_kernel_equals_oracle(blob, k, pool):
    """k_tree_par: parallel_search_num simulations in flight per game on the raz-sched-v1 rounds, with and without pruning."""
    par = load_par_golden()
    g0 = next(g for g in par["games"] if g["resolved_play"]["share_mtcs_info_in_self_play"])
    cfg = config_of(g0)
    cfg.play.parallel_search_num = k
    cfg.play.use_solver_turn = cfg.play.use_solver_turn_in_simulation = 0
    cfg.play.thinking_loop = 1
    eng = EmuEngine(cfg, blob, n_games=2, seed=7, sims_hint=14, nodes_per_game=pool)
    eng.start(40, 14)
    eng.run(chunk=8 if pool else 32)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=k)
    for i in range(2):
        plies, summ = O.selfplay_game(ocfg, blob, 7, 40 + i, 14)
        _same(f"par{k}/{i}", recs[i][0], recs[i][1], plies, summ["winner"])


@pytest.mark.parametrize("k,waves,budget,every", [(1, 8, 6, 1), (2, 9, 3, 3), (3, 64, 17, 2)])
def test_emulated_solver_pool_serves_several_games_from_one_set_of_worker_lanes(golden, blob, k, waves, budget, every, monkeypatch):
    """The end-game solver's pool (csrc/raz_solver_pool.h): five games post their solves - exact at the root, win/loss inside
    simulations - into ONE pool of `waves` worker waves whose lanes take subtrees of whichever game is next in the queue and park
    every search after `budget` iterations; each game must still be the oracle's game, whatever the pool's size or budget, and
    whether the pool gets its round after every tree launch or after every `every`-th one (on the GPU that round then runs beside
    the following launches; here, in program order).  (parallel_search_num 2 and 3 with the solver on were not covered before round 5.)"""
    monkeypatch.setenv("RAZ_SOLVER_BUDGET", str(budget))
    cfg = config_of(_variant(golden, "mini_solver_noresign"))
    cfg.play.parallel_search_num = k
    cfg.play.thinking_loop = 1
    eng = EmuEngine(cfg, blob, n_games=5, seed=43, sims_hint=10, solver_pool_waves=waves, solver_pool_every=every)
    eng.start(70, 10)
    eng.run(chunk=32)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=k)
    solved = 0
    for i in range(5):
        plies, summ = O.selfplay_game(ocfg, blob, 43, 70 + i, 10)
        _same(f"pool/par{k}/{i}", recs[i][0], recs[i][1], plies, summ["winner"])
        solved += sum(p["solved"] for p in plies)
    assert solved > 0


@pytest.mark.parametrize("k,pool,budget", [(4, None, None), (3, 500, 3), (2, None, None)])
def test_emulated_slot_kernel_with_the_solver_equals_oracle(golden, blob, k, pool, budget, monkeypatch):
    """k_tree_par with the end-game solver on: simulations suspended at a solve that ran out of the launch's budget (as new
    simulations in C / C' and as woken sleepers in D) go on at the next launch with the schedule untouched."""
    if budget is not None:
        monkeypatch.setenv("RAZ_SOLVER_BUDGET", str(budget))
    cfg = config_of(_variant(golden, "mini_solver_noresign"))
    cfg.play.parallel_search_num = k
    cfg.play.thinking_loop = 1
    eng = EmuEngine(cfg, blob, n_games=2, seed=43, sims_hint=12, nodes_per_game=pool)
    eng.start(70, 12)
    eng.run(chunk=8 if pool else 32)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=k)
    solved = 0
    for i in range(2):
        plies, summ = O.selfplay_game(ocfg, blob, 43, 70 + i, 12)
        _same(f"par{k}/solver/{i}", recs[i][0], recs[i][1], plies, summ["winner"])
        solved += sum(p["solved"] for p in plies)
    assert solved > 0


def test_emulated_series_on_a_carried_tree_and_position_api(golden, blob):
    """raz_engine_next_game (k_adopt_all: the next game of the slot on the slot's tree) == the oracle's series; then
    raz_engine_set_position + raz_engine_read_node: a search armed at a mid-game position == the oracle taken up there."""
    cfg = config_of(_variant(golden, "mini_shared"))
    cfg.play.thinking_loop = 1
    eng = EmuEngine(cfg, blob, n_games=1, seed=3, nodes_per_game=2 * (2 * (10 * 62 + 128)))
    tree = O.Tree()
    ocfg = O.play_cfg_from_config(cfg)
    for r in range(2):
        (eng.start if r == 0 else eng.next_game)(70 + r, 10)
        eng.run(chunk=32, allow_gc=False)
        (plies, summ), = eng.records(save_policy_of_tau_1=True)
        oplies, osum = O.selfplay_game(ocfg, blob, 3, 70 + r, 10, tree=tree)
        _same(f"series/{r}", plies, summ, oplies, osum["winner"])
    # a game taken up at a position
    start = (int(oplies[20]["own"]), int(oplies[20]["enemy"]), 1) if oplies[20]["player"] == 1 else (int(oplies[20]["enemy"]), int(oplies[20]["own"]), 2)
    eng2 = EmuEngine(cfg, blob, n_games=2, seed=3, sims_hint=10)
    eng2.start(200, 10, n_active=0)
    eng2.set_positions(1, [start[0]], [start[1]], [start[2]], 10, enable_resign=True, one_move=False)   # (the batched form of set_position)
    for _ in range(6):
        eng2.step(1)
    found, w64, n64, _ = eng2.read_node(1, int(oplies[20]["own"]), int(oplies[20]["enemy"]), 1, 0)
    assert found and int(n64.sum()) >= 1
    done = int(n64.sum()) + 1
    fplies, _ = O.selfplay_game(ocfg, blob, 3, 201, done, stop_after_plies=1, start=start)
    assert np.array_equal(np.array(fplies[0]["root_n"]), n64.astype(np.float64))
    assert np.array_equal(np.array(fplies[0]["root_w"]).view(np.uint64), w64.view(np.uint64))


def test_emulated_byte_limited_pool_and_continuous_batching(golden, blob):
    """(1) A pool whose BYTES, not its node count, are the binding limit (compact nodes are variable-size: 40 B + 20 B per legal
    move): pruning is triggered by raz_engine_stats.max_pool_bytes and nothing changes.  (2) raz_engine_harvest: 5 game ids
    through 2 slots (a finished slot restarts on the next id with an empty pool) land in the outbox in id order == the
    oracle's games."""
    from reversi_alpha_zero_amd.engine import raw_from_packed
    cfg = config_of(_variant(golden, "agz"))
    ocfg = O.play_cfg_from_config(cfg)
    eng = EmuEngine(cfg, blob, n_games=2, seed=9, sims_hint=12, nodes_per_game=4096, pool_bytes_per_game=40 * 1024)
    eng.start(300, 12)
    eng.run(chunk=8)
    assert eng.gc_runs >= 2
    for i, (plies, summ) in enumerate(eng.records(save_policy_of_tau_1=False)):
        op, osum = O.selfplay_game(ocfg, blob, 9, 300 + i, 12)
        _same(f"bytes/{i}", plies, summ, op, osum["winner"])
    eng2 = EmuEngine(cfg, blob, n_games=2, seed=9, sims_hint=10, record_root_w=False)
    outbox = eng2.play_continuous(600, 5, 10)
    raw = raw_from_packed(outbox["headers"], outbox["root_n"], outbox["summary"])
    assert list(raw["game_id"]) == [600, 601, 602, 603, 604] and outbox["done"].all()
    for r in range(5):
        op, osum = O.selfplay_game(ocfg, blob, 9, 600 + r, 10)
        n = int(raw["n_plies"][r])
        assert [int(a) for a in raw["headers"][r, :n]["action"]] == [p["action"] for p in op]
        assert all([float(x) for x in raw["root_n"][r, i]] == p["root_n"] for i, p in enumerate(op))
        assert (int(raw["status"][r]) & 0x0f) == osum["winner"]


def test_emulated_deep_search_first_plies(golden, blob):
    """A search 600 simulations deep (deep paths, wide fan-out at the root, a pool that k_gc has to prune within the first plies)
    == the oracle for the first three plies: the regime of BASELINE configs[4] (3200 sims/move; on the GPU: tests/test_config5_gpu.py)."""
    cfg = config_of(_variant(golden, "agz"))
    eng = EmuEngine(cfg, blob, n_games=1, seed=13, nodes_per_game=1500)
    eng.start(77, 600)
    cap, steps = 1500, 0
    while int(eng.read_raw()["n_plies"][0]) < 3:
        eng.step(64)
        steps += 64
        st = eng.stats()
        if eng.pool_nearly_full(st, 64):
            eng.gc(min(cap // 4, st["max_pool_used"] // 2))
        assert steps < 4000
    (plies, _), = eng.records(save_policy_of_tau_1=False)
    oplies, _ = O.selfplay_game(O.play_cfg_from_config(cfg), blob, 13, 77, 600, stop_after_plies=3)
    for i in range(3):
        assert plies[i]["action"] == oplies[i]["action"] and plies[i]["root_n"] == oplies[i]["root_n"] and plies[i]["root_w"] == oplies[i]["root_w"], i
    assert max(oplies[2]["root_n"]) > 100


def test_emulated_leaf_cache_changes_nothing(golden, blob):
    """raz_engine_set_leaf_cache: repeated positions served from the cross-game table (claim / resolve / fill kernels of
    csrc/raz_leaf_cache.hip, here with a table small enough to run out of room) leave every record as it was."""
    cfg = config_of(_variant(golden, "agz"))
    out = []
    for cache in (None, 6):
        eng = EmuEngine(cfg, blob, n_games=4, seed=2, sims_hint=8)
        if cache:
            eng.attach_leaf_cache(10, 0)
        eng.start(10, 8)
        eng.run(chunk=32)
        out.append(eng.records(save_policy_of_tau_1=False))
        if cache:
            st = eng.leaf_cache_stats()
            assert st["hits"] + st["in_batch_duplicates"] > 0 and st["evaluated"] > 0
    assert out[0] == out[1]


@pytest.mark.parametrize("alpha", [1.3, 4.0])
def test_emulated_dirichlet_alpha_above_one(golden, blob, alpha):
    """lib/bitboard.py:162-171 takes any alpha: above 1 numpy's legacy gamma sampler is Marsaglia-Tsang, restated on
    counter-based draws in csrc/raz_detmath.h (attempts evaluated in parallel across lanes) and oracle/orc_rng.c."""
    cfg = config_of(_variant(golden, "mini_shared"))
    cfg.play.dirichlet_alpha, cfg.play.noise_eps, cfg.play.thinking_loop = alpha, 0.4, 1
    eng = EmuEngine(cfg, blob, n_games=2, seed=19, sims_hint=10)
    eng.start(810, 10)
    eng.run(chunk=32)
    ocfg = O.play_cfg_from_config(cfg)
    for i, (plies, summ) in enumerate(eng.records(save_policy_of_tau_1=True)):
        op, osum = O.selfplay_game(ocfg, blob, 19, 810 + i, 10)
        _same(f"alpha{alpha}/{i}", plies, summ, op, osum["winner"])


@pytest.mark.parametrize("budget", [None, 4])
def test_emulated_eval_matches_equal_the_reference_evaluate_games(budget, monkeypatch):
    """(budget: solver iterations per game and launch, see test_emulated_solver_games_equal_oracle - the as-shipped match has 8
    simulations in flight and the solver both at the root and inside them.)
    tests/golden/eval_games.json (the UNMODIFIED reference's EvaluateWorker.play_game, worker/evaluate.py:66-96: best model against
    challenger, two ReversiPlayers with trees and random streams of their own; tests/golden/make_golden_eval.py) replayed the way
    reversi-alpha-zero_amd/worker/evaluate.py plays a match - one engine per model, slot g = that model's player of game g, armed move
    by move with raz_engine_set_position - on the EMULATED tree kernels: every ply's mover, action (resignations included) and root
    visit counts, and the outcome, are the reference's.  (The GPU form of this test drives the product's EvaluateWorker itself:
    tests/test_engine_gpu.py.)"""
    import hashlib
    import json
    import os
    import types
    from conftest import ROOT
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    if budget is not None:
        monkeypatch.setenv("RAZ_SOLVER_BUDGET", str(budget))
    from reversi_alpha_zero_amd.env.reversi_env import Player, ReversiEnv, Winner
    with open(os.path.join(ROOT, "tests", "golden", "eval_games.json")) as f:
        gold = json.load(f)
    blobs = []
    for meta in gold["nets"]:
        b = ReversiNet(meta["filters"], meta["res_layers"], meta["value_fc"]).keras_init_(meta["keras_init_seed"]).randomize_bn_(meta["randomize_bn_seed"]).to_blob()
        assert hashlib.sha256(b).hexdigest() == meta["blob_sha256"]
        blobs.append(b)
    for m in gold["matches"]:
        # the reference reads five settings from config.play whatever play_config says (agent/player.py:127,237,264,410): the product's
        # effective_play_config makes that split, and the match "sections_disagree" pins it
        from reversi_alpha_zero_amd.agent.player import effective_play_config
        whole = types.SimpleNamespace(play=types.SimpleNamespace(**m["resolved_config_play"]))
        pc = effective_play_config(whole, types.SimpleNamespace(**m["resolved_play_config"]))
        if m["name"] == "eval_par4_sections_disagree":
            assert pc.virtual_loss == 2 and pc.use_solver_turn_in_simulation == 48 and pc.use_solver_turn == 52
        cfg = types.SimpleNamespace(play=pc, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
        n, first, sims = len(m["games"]), m["first_game_id"], int(pc.simulation_num_per_move)
        sides = {}
        for is_best, blob, seed in ((True, blobs[0], 2 * m["seed"]), (False, blobs[1], 2 * m["seed"] + 1)):
            e = EmuEngine(cfg, blob, n, seed=seed, sims_hint=sims, max_plies=64)
            e.start(first, sims, n_active=0)
            sides[is_best] = e
        envs = [ReversiEnv().reset() for _ in range(n)]
        best_is_black = [g["best_is_black"] for g in m["games"]]
        plies = [[] for _ in range(n)]
        live = list(range(n))
        while live:
            armed = {True: [], False: []}
            for g in live:
                own, enemy = envs[g].get_own_and_enemy()
                mover_is_best = (envs[g].next_player == Player.black) == best_is_black[g]
                sides[mover_is_best].set_position(g, own, enemy, 1, sims, enable_resign=True, one_move=True)
                armed[mover_is_best].append(g)
            for is_best, e in sides.items():
                if not armed[is_best]:
                    continue
                while e.stats()["idle_or_done"] < n:
                    e.step(16)
                raw = e.read_raw()
                for g in armed[is_best]:
                    a = int(raw["headers"][g, 0]["action"])
                    plies[g].append(("best" if is_best else "ng", a, [float(x) for x in raw["root_n"][g, 0]]))
                    envs[g].step(a if a >= 0 else None)
            live = [g for g in live if not envs[g].done]
        for g, ref in enumerate(m["games"]):
            where = (m["name"], ref["game_id"])
            for i, ((who, a, rn), p) in enumerate(zip(plies[g], ref["plies"])):
                assert (who, a) == (p["who"], p["action"]), (where, i, (who, a), (p["who"], p["action"]))
                if p["root_n"] is not None:
                    want = [0.0] * 64
                    for k, v in p["root_n"].items():
                        want[int(k)] = v
                    assert rn == want, (where, i)
            assert len(plies[g]) == len(ref["plies"]), where
            env = envs[g]
            ng_win = None if env.winner not in (Winner.black, Winner.white) else int((env.winner == Winner.black) != best_is_black[g])
            assert ng_win == ref["ng_win"] and list(env.observation.number_of_black_and_white) == ref["black_white"], where


@pytest.mark.parametrize("budget", [None, 7])
def test_emulated_device_solver_equals_compiled_cython(budget, monkeypatch):
    """(budget: iterations a root solve may run per launch before it is parked in the game's workspace - the product's 384, and 7:
    hundreds of park / resume cycles per solve, on the positions of at most 9 empties.)
    tests/golden/solver_kat.json (answers of the reference's COMPILED Cython solver: its three known answers and 120 late-game
    positions) through the device solver on the emulator, exact mode at the root (agent/player.py:100-103,150-161): positions of
    7..14 empties run the lane-parallel search (csrc/raz_engine_core.h solver_solve: the root's moves and their replies expanded into
    tasks, one reference-shaped depth-first search per lane), smaller ones the scalar one; move and sign(score) must be the Cython
    solver's.  (GPU form: tests/test_oracle_solver.py.)"""
    import json
    import os
    import types
    from conftest import ROOT
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    with open(os.path.join(ROOT, "tests", "golden", "solver_kat.json")) as f:
        kat = json.load(f)
    play = types.SimpleNamespace(
        simulation_num_per_move=8, share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=40,
        start_rethinking_turn=10, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=10, virtual_loss=3,
        parallel_search_num=1, resign_threshold=None, allowed_resign_turn=10, disable_resignation_rate=0.0,
        use_solver_turn=46, use_solver_turn_in_simulation=46)
    cfg = types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
    cases = [(int(k["black"], 16), int(k["white"], 16), k["next_player"], k["answer"] if k["exactly"] else k["answer_other_mode"]) for k in kat["kat"]]
    cases += [(int(p["black"], 16), int(p["white"], 16), p["next_player"], p["exact"]) for p in kat["positions"]]
    cases = [c for c in cases if bin(c[0] | c[1]).count("1") - 4 >= 46]
    empties = sorted({64 - bin(c[0] | c[1]).count("1") for c in cases})
    assert len(cases) >= 100 and empties[0] <= 6 and empties[-1] >= 10, empties     # both searches are exercised
    if budget is not None:
        monkeypatch.setenv("RAZ_SOLVER_BUDGET", str(budget))
        cases = [c for c in cases if 7 <= 64 - bin(c[0] | c[1]).count("1") <= 9]
        assert len(cases) >= 30
    eng = EmuEngine(cfg, ReversiNet(16, 1, 16).keras_init_(0).to_blob(), len(cases), seed=3, sims_hint=8)
    eng.start(0, 8)
    for g, (b, w, pl, _) in enumerate(cases):
        eng.set_position(g, b, w, pl, 8, enable_resign=False, one_move=True)
    for _ in range(40000):      # a 10-empties exact solve takes many launches: it runs on a per-launch budget and is parked in between
        eng.step(4)
        if eng.stats()["idle_or_done"] >= len(cases):
            break
    raw = eng.read_raw()
    for g, (b, w, pl, (move, score)) in enumerate(cases):
        h = raw["headers"][g, 0]
        assert int(raw["n_plies"][g]) == 1 and int(h["flags"]) & 1, (g, "not solved")
        assert int(h["action"]) == move and float(h["n"]) == 999.0 and float(h["q"]) == float(np.sign(score)), (g, int(h["action"]), move, float(h["q"]), score)


@pytest.mark.parametrize("par,budget", [(1, None), (4, 40)])
def test_emulated_in_simulation_solves_of_11_to_13_empties_equal_oracle(golden, blob, par, budget, monkeypatch):
    """The lane-parallel solver on LARGE task trees (three plies below a position of 11-13 empties: up to ~150 level-2 nodes - three
    rounds of the 64 lanes - and ~1500 tasks), non-exact mode: positions taken up by raz_engine_set_position with the in-simulation
    solver from turn 46 (the declared limit) and no solver at the root, three simulations - the first solves the root position
    itself, the next ones its children - one simulation in flight and four (with a budget of 40 iterations per launch: the 7.5 KB
    task tree is parked and restored many times).  The decided move and the root statistics == the oracle's for the same id from
    the same position."""
    import random
    from reversi_alpha_zero_amd.lib import bitboard as bb
    if budget is not None:
        monkeypatch.setenv("RAZ_SOLVER_BUDGET", str(budget))
    cfg = config_of(_variant(golden, "mini_solver_noresign"))
    cfg.play.use_solver_turn, cfg.play.use_solver_turn_in_simulation = 0, 46
    cfg.play.parallel_search_num, cfg.play.thinking_loop = par, 1
    rng = random.Random(2024)
    starts = []
    while len(starts) < 6:
        b, w, p = 0x0000000810000000, 0x0000001008000000, 1
        want = 64 - (11 + len(starts) % 3)     # discs on the board: 11, 12, 13 empties in turn
        while bin(b | w).count("1") < want:
            own, enemy = (b, w) if p == 1 else (w, b)
            legal = int(bb.find_correct_moves(own, enemy))
            if not legal:
                if not int(bb.find_correct_moves(enemy, own)):
                    break
                p = 3 - p
                continue
            a = rng.choice([i for i in range(64) if legal >> i & 1])
            fl = int(bb.calc_flip(a, own, enemy))
            own, enemy = (own ^ fl) | (1 << a), enemy ^ fl
            b, w = (own, enemy) if p == 1 else (enemy, own)
            p = 3 - p
        own, enemy = (b, w) if p == 1 else (w, b)
        if bin(b | w).count("1") == want and int(bb.find_correct_moves(own, enemy)):
            starts.append((b, w, p))
    sims = 3
    eng = EmuEngine(cfg, blob, n_games=len(starts), seed=9, sims_hint=sims)
    eng.start(300, sims)
    for g, (b, w, p) in enumerate(starts):
        eng.set_position(g, b, w, p, sims, enable_resign=False, one_move=True)
    for _ in range(200000):
        eng.step(8)
        if eng.stats()["idle_or_done"] >= len(starts):
            break
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=par) if par > 1 else O.play_cfg_from_config(cfg)
    cfg.play.use_solver_turn_in_simulation = 0
    ocfg_off = O.play_cfg_from_config(cfg, parallel_search_num=par) if par > 1 else O.play_cfg_from_config(cfg)
    mattered = 0
    for g, start in enumerate(starts):
        oplies, _ = O.selfplay_game(ocfg, blob, 9, 300 + g, sims, stop_after_plies=1, start=start)
        a, ref = recs[g][0][0], oplies[0]
        for key in ("player", "own", "enemy", "action", "root_n", "root_w", "n", "q"):
            assert a[key] == ref[key], (g, key, a[key], ref[key])
        off, _ = O.selfplay_game(ocfg_off, blob, 9, 300 + g, sims, stop_after_plies=1, start=start)
        mattered += off[0]["root_w"] != ref["root_w"] or off[0]["root_n"] != ref["root_n"]
    assert mattered >= 3      # (the solves decided what the simulations returned: without them the statistics differ)


def test_emulated_solver_with_the_last_two_squares_finished_in_the_move_function_in_a_process_of_its_own():
    """RAZ_SOLVER_INLINE_LAST=2 (csrc/raz_solver_pool.h solver_last_two: positions with TWO empty squares finished inside solver_play,
    the reference's loop over the moves with its non-exact stop) is an option the library is not built with by default (level 1 is):
    the device solver against the compiled reference solver and the pool serving several games, on a variant build of the emulated
    kernels in a child process."""
    import os
    import subprocess
    import sys
    if os.environ.get("RAZ_EMU_VARIANT"):
        pytest.skip("a child of this test")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-p", "no:xdist",
                        "-k", "device_solver_equals_compiled_cython or solver_pool_serves_several_games"],
                       env={**os.environ, "RAZ_EMU_VARIANT": "last2", "RAZ_EMU_EXTRA": "-DRAZ_SOLVER_INLINE_LAST=2"},
                       capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]

