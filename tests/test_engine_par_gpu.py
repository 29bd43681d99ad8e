"""GPU: parallel_search_num > 1 (k_tree_par, the asyncio event loop of agent/player.py:189-355 in exact
virtual time = raz-sched-v1) against the CPU oracle and against games of the UNMODIFIED reference run
on ref_harness.VirtualTimeLoop (tests/golden/mcts_par_games.json) - bit-exact, like the one-in-flight
path; plus the slot kernel forced onto parallel_search_num = 1, which must reproduce the ordinary
goldens (the slot machinery itself changes nothing)."""
import numpy as np
import pytest

import oracle as O
from oracle_util import load_mcts_golden, load_par_golden, orc_cfg_of, golden_net_blob, config_of, dense
from test_engine_gpu import _compare_game

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def blob():
    return golden_net_blob(load_mcts_golden()["net"])


def _ref_plies(g):
    return [dict(p, own=int(p["own"], 16), enemy=int(p["enemy"], 16), root_n=dense(p["root_n"]),
                 root_w=dense(p["root_w"]), saved_policy=dense(p["saved_policy"]) if p["has_row"] else None)
            for p in g["plies"]]


def _replay(g, blob, **kw):
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    cfg = config_of(g)
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=1, seed=g["seed"], sims_hint=g["sims_per_move"],
                         record_root_w=True, **kw)
    eng.start(first_game_id=g["game_id"], sims_per_move=g["sims_per_move"])
    st = eng.run(chunk=256)
    (plies, summ), = eng.records(save_policy_of_tau_1=g["resolved_play_data"]["save_policy_of_tau_1"])
    tag = f'{g["variant"]}/{g["game_id"]}'
    _compare_game(tag, plies, summ, _ref_plies(g), g["winner"])
    assert (bool(summ["resigned_black"]), bool(summ["resigned_white"])) == (g["resigned_black"], g["resigned_white"]), tag
    assert (summ["black"], summ["white"]) == (int(g["black"], 16), int(g["white"], 16)), tag
    assert st["nn_leaves"] == g["nn_positions"], tag
    return st


def test_slot_kernel_with_one_slot_reproduces_reference_goldens(blob):
    """parallel_search_num = 1 driven by k_tree_par (reserved bit 3): every golden game of the unmodified
    reference, solver-on and re-thinking variants included."""
    for g in load_mcts_golden()["games"]:
        _replay(g, blob, force_slot_kernel=True)


def test_engine_parallel_search_reproduces_reference_on_virtual_time_loop(blob):
    """parallel_search_num 2, 3, 4, 8, 16 - shared and unshared trees, solver on, re-thinking loops,
    resignation - each game alone in its slot == the reference's record, every root N and W."""
    par = load_par_golden()
    ks = set()
    for g in par["games"]:
        assert g["resolved_play"]["parallel_search_num"] > 1
        ks.add(g["resolved_play"]["parallel_search_num"])
        _replay(g, blob)
    assert ks >= {2, 3, 4, 8, 16}


@pytest.mark.parametrize("variant,k,inner", [("mini_par4_as_shipped", 4, 0), ("agz_par8_unshared", 8, 1),
                                             ("ch5_par8_cpuct5", 5, 15)])
def test_engine_parallel_search_batch_vs_oracle(blob, variant, k, inner):
    """64 concurrent games with mixed simulation counts == 64 independent oracle games: independent of
    batching, of the slices/streams the batch is stepped in and of the per-launch start budget."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in load_par_golden()["games"] if g["variant"] == variant)
    cfg = config_of(g0)
    cfg.play.parallel_search_num = k
    n = 64
    sims = np.array([10 + (i % 5) * 6 for i in range(n)], dtype=np.uint32)
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=n, seed=55, sims_hint=int(sims.max()), record_root_w=True,
                         inner_max=inner)
    eng.start(first_game_id=3000, sims_per_move=sims)
    st = eng.run(chunk=64)
    recs = eng.records(save_policy_of_tau_1=g0["resolved_play_data"]["save_policy_of_tau_1"])
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=k)
    for i in range(0, n, 3):
        plies, summ = O.selfplay_game(ocfg, blob, 55, 3000 + i, int(sims[i]))
        _compare_game(f"{variant}/k{k}/{3000 + i}", recs[i][0], recs[i][1], plies, summ["winner"])
        assert sum(p["sims"] for p in recs[i][0]) == summ["n_sims"]
    assert st["total_sims"] == sum(sum(p["sims"] for p in r[0]) for r in recs)


@pytest.mark.parametrize("variant,pool", [("mini_par2_nosolver", 1024), ("agz_par8_unshared", 768)])
def test_engine_parallel_search_with_node_pruning_equals_oracle(blob, variant, pool):
    """k_gc with simulations in flight in several slots (paths, leaves, slept-on nodes renumbered):
    pools far too small for whole games, 32 games == the oracle."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in load_par_golden()["games"] if g["variant"] == variant)
    cfg = config_of(g0)
    k = g0["resolved_play"]["parallel_search_num"]
    n, sims = 32, 24
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=n, seed=61, nodes_per_game=pool, record_root_w=True)
    eng.start(first_game_id=700, sims_per_move=sims)
    eng.run(chunk=16)
    assert eng.gc_runs >= 2
    recs = eng.records(save_policy_of_tau_1=g0["resolved_play_data"]["save_policy_of_tau_1"])
    ocfg = orc_cfg_of(g0)
    for i in range(0, n, 2):
        plies, summ = O.selfplay_game(ocfg, blob, 61, 700 + i, sims)
        _compare_game(f"gc/{variant}/k{k}/{700 + i}", recs[i][0], recs[i][1], plies, summ["winner"])
