"""CPU: the FUSED tree + net kernel (csrc/raz_engine_fused.hip k_tree_net: every game's wave evaluates its own leaves with
raz_net16_forward_in_wave, `iters` simulations per launch) on the wave emulator - the whole product compiled for the host
(tests/native/libraz_emu_full.so: the tree kernels AND the net kernels on emulated matrix cores) - against the CPU oracle and the
unmodified reference's golden games, bit for bit.  The GPU counterparts are in tests/test_engine_gpu.py."""
import numpy as np
import pytest

import oracle as O
from emu_util import EmuEngine
from oracle_util import load_mcts_golden, golden_net_blob, config_of, dense
from test_engine_emu import _same, _variant


@pytest.fixture(scope="module")
def golden():
    return load_mcts_golden()


@pytest.fixture(scope="module")
def blob(golden):
    return golden_net_blob(golden["net"])


def test_fused_kernel_replays_a_reference_golden_game(golden, blob):
    """A golden game of the unmodified reference (shared tree with re-thinking loops), its leaves evaluated inside the tree
    kernel's wave; launches of 64 simulation steps (two k_tree_net launches of 32 iterations each)."""
    g = next(g for g in golden["games"] if g["variant"] == "mini_shared" and g["sims_per_move"] <= 40)
    cfg = config_of(g)
    eng = EmuEngine(cfg, blob, n_games=1, seed=g["seed"], sims_hint=g["sims_per_move"], fused=True)
    eng.start(g["game_id"], g["sims_per_move"])
    eng.run(chunk=64)
    (plies, summ), = eng.records(save_policy_of_tau_1=g["resolved_play_data"]["save_policy_of_tau_1"])
    ref = [dict(p, own=int(p["own"], 16), enemy=int(p["enemy"], 16), root_n=dense(p["root_n"]), root_w=dense(p["root_w"])) for p in g["plies"]]
    _same("fused/" + g["variant"], plies, summ, ref, g["winner"])
    assert (summ["black"], summ["white"]) == (int(g["black"], 16), int(g["white"], 16))


@pytest.mark.parametrize("variant,pool,chunk", [("agz", 200, 5)])
def test_fused_batch_equals_oracle_with_pruning_and_partial_launches(golden, blob, variant, pool, chunk):
    """A small batch with mixed simulation counts == independent oracle games; launches that end in the middle of a search
    (5 iterations: the last leaf's answer waits in the leaf exchange for the next launch) and pools pruned between launches."""
    cfg = config_of(_variant(golden, variant))
    n = 3
    sims = np.array([9, 12, 15], dtype=np.uint32)
    eng = EmuEngine(cfg, blob, n_games=n, seed=31, sims_hint=15, nodes_per_game=pool, fused=True)
    eng.start(500, sims)
    eng.run(chunk=chunk)
    assert pool is None or eng.gc_runs >= 2
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg)
    for i in range(n):
        plies, summ = O.selfplay_game(ocfg, blob, 31, 500 + i, int(sims[i]))
        _same(f"fused/{variant}/{pool}/{i}", recs[i][0], recs[i][1], plies, summ["winner"])


def test_fused_kernel_is_refused_where_it_does_not_apply(golden, blob):
    """Nets that are not 16 filters wide: raz_engine_create says so instead of running something else."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    cfg = config_of(_variant(golden, "mini_shared"))
    with pytest.raises(RuntimeError, match="fused"):
        EmuEngine(cfg, ReversiNet(32, 1, 16).keras_init_(0).to_blob(), n_games=1, sims_hint=8, fused=True)


@pytest.mark.parametrize("k,pool", [(4, None), (3, 400)])
def test_fused_slot_kernel_equals_oracle(blob, k, pool):
    """k_tree_par_net: parallel_search_num simulations in flight per game on the raz-sched-v1 rounds, every round's queued leaves
    evaluated by the game's own wave before the next round; with and without pruning."""
    from oracle_util import load_par_golden
    par = load_par_golden()
    g0 = next(g for g in par["games"] if g["resolved_play"]["share_mtcs_info_in_self_play"])
    cfg = config_of(g0)
    cfg.play.parallel_search_num = k
    cfg.play.use_solver_turn = cfg.play.use_solver_turn_in_simulation = 0
    cfg.play.thinking_loop = 1
    eng = EmuEngine(cfg, blob, n_games=2, seed=7, sims_hint=14, nodes_per_game=pool, fused=True)
    eng.start(40, 14)
    eng.run(chunk=8 if pool else 32)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=k)
    for i in range(2):
        plies, summ = O.selfplay_game(ocfg, blob, 7, 40 + i, 14)
        _same(f"fused/par{k}/{i}", recs[i][0], recs[i][1], plies, summ["winner"])


@pytest.mark.parametrize("k,budget", [(1, None), (1, 5), (3, 4)])
def test_fused_solver_game_equals_oracle(golden, blob, k, budget, monkeypatch):
    """mini.yml as shipped (exact solver at the root, win/loss solver inside simulations): the SOLVER = true forms of k_tree_net and
    k_tree_par_net.  budget: solver iterations per game and launch - solves that run out are suspended (the root's in begin_move,
    the others in the middle of their descent) and go on at the next launch."""
    if budget is not None:
        monkeypatch.setenv("RAZ_SOLVER_BUDGET", str(budget))
    cfg = config_of(_variant(golden, "mini_solver_noresign"))
    cfg.play.parallel_search_num = k
    eng = EmuEngine(cfg, blob, n_games=1, seed=41, sims_hint=10, fused=True)
    eng.start(900, 10)
    eng.run(chunk=32 if budget is None else 1)   # (a launch is over for a game whose solve is suspended, however many iterations it had left)
    (plies, summ), = eng.records(save_policy_of_tau_1=True)
    oplies, osum = O.selfplay_game(O.play_cfg_from_config(cfg, parallel_search_num=k), blob, 41, 900, 10)
    _same("fused/solver", plies, summ, oplies, osum["winner"])
    assert sum(p["solved"] for p in oplies) > 0


def test_fused_series_position_api_and_continuous_batching(golden, blob):
    """The host-side flows around the kernel are unchanged by it: raz_engine_next_game (the next game of a slot on the slot's tree),
    a search armed at a mid-game position and stepped ONE iteration per launch (raz_engine_set_positions / raz_engine_read_node),
    and raz_engine_harvest (5 game ids through 2 slots) == the oracle."""
    from reversi_alpha_zero_amd.engine import raw_from_packed
    cfg = config_of(_variant(golden, "mini_shared"))
    cfg.play.thinking_loop = 1
    eng = EmuEngine(cfg, blob, n_games=1, seed=3, nodes_per_game=2 * (2 * (10 * 62 + 128)), fused=True)
    tree = O.Tree()
    ocfg = O.play_cfg_from_config(cfg)
    for r in range(2):
        (eng.start if r == 0 else eng.next_game)(70 + r, 10)
        eng.run(chunk=32, allow_gc=False)
        (plies, summ), = eng.records(save_policy_of_tau_1=True)
        oplies, osum = O.selfplay_game(ocfg, blob, 3, 70 + r, 10, tree=tree)
        _same(f"fused/series/{r}", plies, summ, oplies, osum["winner"])
    start = (int(oplies[20]["own"]), int(oplies[20]["enemy"]), 1) if oplies[20]["player"] == 1 else (int(oplies[20]["enemy"]), int(oplies[20]["own"]), 2)
    eng2 = EmuEngine(cfg, blob, n_games=2, seed=3, sims_hint=10, fused=True)
    eng2.start(200, 10, n_active=0)
    eng2.set_positions(1, [start[0]], [start[1]], [start[2]], 10, enable_resign=True, one_move=False)
    for _ in range(6):
        eng2.step(1)
    found, w64, n64, _ = eng2.read_node(1, int(oplies[20]["own"]), int(oplies[20]["enemy"]), 1, 0)
    assert found and int(n64.sum()) >= 1
    fplies, _ = O.selfplay_game(ocfg, blob, 3, 201, int(n64.sum()) + 1, stop_after_plies=1, start=start)
    assert np.array_equal(np.array(fplies[0]["root_n"]), n64.astype(np.float64))
    assert np.array_equal(np.array(fplies[0]["root_w"]).view(np.uint64), w64.view(np.uint64))
    cfg3 = config_of(_variant(golden, "agz"))
    ocfg3 = O.play_cfg_from_config(cfg3)
    eng3 = EmuEngine(cfg3, blob, n_games=2, seed=9, sims_hint=10, record_root_w=False, fused=True)
    outbox = eng3.play_continuous(600, 5, 10)
    raw = raw_from_packed(outbox["headers"], outbox["root_n"], outbox["summary"])
    assert list(raw["game_id"]) == [600, 601, 602, 603, 604] and outbox["done"].all()
    for r in range(5):
        op, osum = O.selfplay_game(ocfg3, blob, 9, 600 + r, 10)
        n = int(raw["n_plies"][r])
        assert [int(a) for a in raw["headers"][r, :n]["action"]] == [p["action"] for p in op]
        assert all([float(x) for x in raw["root_n"][r, i]] == p["root_n"] for i, p in enumerate(op))
        assert (int(raw["status"][r]) & 0x0f) == osum["winner"]
