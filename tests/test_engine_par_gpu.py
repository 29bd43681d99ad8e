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


def test_reversi_player_facade_parallel_search_equals_oracle(blob):
    """A whole game through the ReversiPlayer drop-in with mini.yml's play settings as shipped
    (parallel_search_num 4, thinking_loop 2, solver from turn 50) == the oracle's game at the same setting:
    actions, n, q, resign flags and the training rows black.moves + white.moves."""
    import json
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from oracle_util import rows_of_game
    from test_engine_gpu import _facade_game
    par = load_par_golden()
    g0 = next(g for g in par["games"] if g["variant"] == "mini_par4_as_shipped")
    cfg = Config()
    cfg.play.update(g0["resolved_play"])
    cfg.play_data.update(g0["resolved_play_data"])
    assert cfg.play.parallel_search_num == 4
    meta = par["net"]
    net = ReversiNet(meta["filters"], meta["res_layers"], meta["value_fc"]).keras_init_(meta["keras_init_seed"])
    net.randomize_bn_(meta["randomize_bn_seed"])
    seed, gid, sims = 6, 41, 18
    env, black, white, evals = _facade_game(cfg, net, seed, gid, sims)
    plies, summ = O.selfplay_game(O.play_cfg_from_config(cfg, parallel_search_num=4), blob, seed, gid, sims)
    assert len(evals) == len(plies)
    for (pl, ae), p in zip(evals, plies):
        assert pl == p["player"]
        assert (ae.action if ae.action is not None else -1) == p["action"]
        if ae.action is not None:
            assert float(ae.n) == p["n"] and float(ae.q) == p["q"]
    assert {"black": 1, "white": 2, "draw": 3}[env.winner.name] == summ["winner"]
    assert (black.resigned, white.resigned) == (bool(summ["resigned_black"]), bool(summ["resigned_white"]))
    assert json.dumps(black.moves + white.moves) == json.dumps(rows_of_game(plies, summ["winner"]))


def test_reversi_player_stop_thinking_with_simulations_in_flight(blob):
    """stop_thinking() at parallel_search_num 8: simulations already started return (their virtual losses
    are taken back), no new one starts, and the move is decided from the tree as it is."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.agent.player import ReversiPlayer, CallbackInMCTS
    from reversi_alpha_zero_amd.lib.bitboard import find_correct_moves
    from reversi_alpha_zero_amd.env.reversi_env import ReversiEnv, Player
    par = load_par_golden()
    g0 = next(g for g in par["games"] if g["variant"] == "agz_par8_unshared")
    cfg = Config()
    cfg.play.update(g0["resolved_play"])
    cfg.play_data.update(g0["resolved_play_data"])
    cfg.play.simulation_num_per_move = 400
    meta = par["net"]
    net = ReversiNet(meta["filters"], meta["res_layers"], meta["value_fc"]).keras_init_(meta["keras_init_seed"])
    p = ReversiPlayer(cfg, net, enable_resign=False)
    calls = []

    def cb(q, n):
        calls.append(sum(n))
        if len(calls) == 3:
            p.stop_thinking()

    env = ReversiEnv().reset()
    env.step(19)
    own, enemy = env.board.white, env.board.black
    ae = p.action_with_evaluation(own, enemy, callback_in_mtcs=CallbackInMCTS(5, cb))
    assert ae.action is not None and (find_correct_moves(own, enemy) >> ae.action) & 1
    key = ReversiPlayer.counter_key(ReversiEnv().update(own, enemy, Player.black))
    n, w = p.var_n[key], p.var_w[key]
    assert len(calls) >= 3 and 0 < sum(n) < 400
    assert float(sum(n)) == float(int(sum(n))) and abs(float(ae.q)) <= 1.0   # no virtual loss left behind
    assert all(abs(wi) <= ni + 1e-9 for wi, ni in zip(w, n))


# ---- reset_mtcs_info_per_game > 1: the slot's tree carried from game to game (raz_engine_next_game) ----------
def test_engine_tree_carried_across_games_reproduces_reference_series(blob):
    """config/mini.yml as shipped (reset_mtcs_info_per_game 3, parallel_search_num 4, thinking_loop 2, solver from
    turn 50) and the same at parallel_search_num 1: three consecutive games of one reference worker on one
    MCTSInfo == three raz_engine_next_game rounds of one slot, every root N and W."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    ser = load_mcts_golden("mcts_series_games.json")
    dnet = DeviceNet(blob, DEV)
    eng, cur = None, None
    for g in ser["games"]:
        cfg = config_of(g)
        if g["variant"] != cur:
            assert g["series_index"] == 0
            cur = g["variant"]
            eng = SelfPlayEngine(cfg, dnet, n_games=1, seed=g["seed"], record_root_w=True,
                                 nodes_per_game=3 * (g["sims_per_move"] * cfg.play.thinking_loop * 62 + 128) * 2)
            eng.start(first_game_id=g["game_id"], sims_per_move=g["sims_per_move"])
        else:
            eng.next_game(first_game_id=g["game_id"], sims_per_move=g["sims_per_move"])
        st = eng.run(chunk=256, allow_gc=False)
        (plies, summ), = eng.records(save_policy_of_tau_1=g["resolved_play_data"]["save_policy_of_tau_1"])
        tag = f'{g["variant"]}/{g["game_id"]}'
        _compare_game(tag, plies, summ, _ref_plies(g), g["winner"])
        assert st["nn_leaves"] == g["nn_positions"], tag
        assert (bool(summ["resigned_black"]), bool(summ["resigned_white"])) == (g["resigned_black"], g["resigned_white"]), tag


def test_worker_series_of_three_rounds_equals_oracle(blob, tmp_path):
    """The `self` worker with mini.yml's play settings as shipped: three play_batch rounds of 12 slots = 12 series
    of 3 games on carried trees; every game == the oracle's game on the series' tree, and the next round (a new
    series) starts from empty trees again."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    g0 = next(g for g in load_mcts_golden("mcts_series_games.json")["games"] if g["variant"] == "mini_yml_as_shipped_3_games")
    cfg = Config()
    cfg.play.update(g0["resolved_play"])
    cfg.play.schedule_of_simulation_num_per_move = [(0, 10)]
    cfg.play_data.update(g0["resolved_play_data"])
    rc = cfg.resource
    rc.data_dir = str(tmp_path)
    rc.force_simulation_num_file = str(tmp_path / ".force-sim")
    n = 12
    w = BatchedSelfPlayWorker(cfg, blob, games_in_flight=n, seed=9, device=DEV)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=cfg.play.parallel_search_num)
    trees = [O.Tree() for _ in range(n)]
    for rnd in range(4):
        if rnd == 3:
            trees = [O.Tree() for _ in range(n)]   # reset_mtcs_info_per_game = 3: the 4th round starts a new series
        recs = w.play_batch(first_game_idx=100 + rnd * n)
        for i in range(0, n, 2 if rnd else 1):
            plies, summ = O.selfplay_game(ocfg, blob, 9, 100 + rnd * n + i, 10, tree=trees[i])
            _compare_game(f"series/round{rnd}/slot{i}", recs[i][0], recs[i][1], plies, summ["winner"], check_w=False)
        if rnd:   # the slots skipped above still have to advance their oracle trees
            for i in range(1, n, 2):
                O.selfplay_game(ocfg, blob, 9, 100 + rnd * n + i, 10, tree=trees[i])


def test_self_play_start_entry_point_writes_the_series_files(blob, tmp_path, monkeypatch):
    """worker.self_play.start(config) - the reference's entry point (worker/self_play.py:28) - with mini.yml's play
    settings as shipped: 3 rounds of 8 slots (one series of 3 games per slot on a carried tree), through the
    production path (record arrays -> native JSON text).  The files' rows == the oracle's games in file order, the
    game-index file is advanced, and they load the way the reference's trainer reads them."""
    import json
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker import self_play
    from reversi_alpha_zero_amd.lib.data_helper import get_game_data_filenames, read_game_data_from_file
    from oracle_util import rows_of_game
    g0 = next(g for g in load_mcts_golden("mcts_series_games.json")["games"] if g["variant"] == "mini_yml_as_shipped_3_games")
    cfg = Config()
    cfg.play.update(g0["resolved_play"])
    cfg.play.schedule_of_simulation_num_per_move = [(0, 8)]
    cfg.play_data.update(dict(g0["resolved_play_data"], nb_game_in_file=4, nb_game_in_ggf_file=6))
    rc = cfg.resource
    rc.data_dir = str(tmp_path)
    rc.play_data_dir = str(tmp_path / "play_data")
    rc.self_play_ggf_data_dir = str(tmp_path / "ggf")
    rc.model_dir = str(tmp_path / "model")
    rc.next_generation_model_dir = str(tmp_path / "model" / "next")
    rc.log_dir = str(tmp_path / "logs")
    rc.project_dir = str(tmp_path)
    rc.force_simulation_num_file = str(tmp_path / ".force-sim")
    rc.self_play_game_idx_file = str(tmp_path / ".self-play-game-idx")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    n = 8
    w = self_play.start(cfg, net_blob=blob, games_in_flight=n, total_games=3 * n, seed=13)
    assert open(rc.self_play_game_idx_file).read() == str(3 * n)
    got = [row for f in get_game_data_filenames(rc) for row in read_game_data_from_file(f)]
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=cfg.play.parallel_search_num)
    trees = [O.Tree() for _ in range(n)]
    exp = []
    for rnd in range(3):
        for i in range(n):
            plies, summ = O.selfplay_game(ocfg, blob, 13, rnd * n + i, 8, tree=trees[i])
            exp += rows_of_game(plies, summ["winner"])
    assert json.dumps(got) == json.dumps(exp)
    assert len(get_game_data_filenames(rc)) == 6 and len(list((tmp_path / "ggf").iterdir())) >= 4
