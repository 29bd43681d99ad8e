"""The reference's OWN consumers, unmodified, on drop-in output (SURVEY §8(d) Config 1: "assert loadable by
OptimizeWorker.convert_to_training_data").  CPU, `needs_reference`:

  * one block of real games (played by the CPU oracle - test infrastructure) goes through BatchedSelfPlayWorker.run() - gather,
    resignation bookkeeping, the native row emitter, the background writer - into play_*.json files; then the reference's
    lib/data_helper.get_game_data_filenames + worker/optimize.py OptimizeWorker.load_play_data / convert_to_training_data
    (optimize.py:165-231) read that directory and must see exactly the rows of those games;
  * the reference's EvaluateWorker.play_game (evaluate.py:66-95) plays a game between two drop-in ReversiPlayer objects
    (agent/player.py of this package) through the reference's own ReversiEnv; the players' engine is the wave-emulator build of
    the tree kernels here (there is no GPU in this container; on a GPU box the same class runs on libraz.so).
"""
import os
import random

import numpy as np
import pytest

import oracle as O
from oracle_util import load_mcts_golden, golden_net_blob, config_of

pytestmark = pytest.mark.needs_reference


def _oracle_games(first_id, n, sims=6, seed=11):
    golden = load_mcts_golden()
    blob = golden_net_blob(golden["net"])
    cfg = config_of(next(g for g in golden["games"] if g["variant"] == "mini_shared"))
    cfg.play.thinking_loop = 1
    ocfg = O.play_cfg_from_config(cfg)
    games = []
    for gid in range(first_id, first_id + n):
        plies, summ = O.selfplay_game(ocfg, blob, seed, gid, sims)
        summ = dict(summ, game_id=gid, status=summ["winner"])
        games.append((plies, summ))
    return cfg, games


def test_reference_optimize_worker_loads_the_files_the_drop_in_worker_writes(tmp_path, monkeypatch):
    import test_worker_run_host as H
    ocfg_like, games = _oracle_games(0, 6)
    monkeypatch.setattr(H, "stub_games", lambda first, n, seed=7: games[first:first + n])
    cfg = H.make_config(tmp_path)
    cfg.play_data.update(dict(nb_game_in_file=2, nb_game_in_ggf_file=100, drop_draw_game_rate=0.0, max_file_num=1000,
                              save_policy_of_tau_1=True))
    cfg.play.change_tau_turn = ocfg_like.play.change_tau_turn
    cfg.play.resign_threshold = None
    w = H.make_stub_worker(cfg, games_in_flight=6)
    w.run(total_games=6, background_emit=True)
    files = sorted(os.listdir(cfg.resource.play_data_dir))
    assert len(files) == 3

    # ---- the unmodified reference reads the directory
    import ref_harness as rh
    rh.install()
    from reversi_zero.lib.data_helper import get_game_data_filenames, read_game_data_from_file
    from reversi_zero.worker.optimize import OptimizeWorker
    rcfg = rh.load_config("mini.yml")
    rcfg.resource.play_data_dir = cfg.resource.play_data_dir
    names = get_game_data_filenames(rcfg.resource)
    assert [os.path.basename(p) for p in names] == files
    ow = OptimizeWorker(rcfg)
    ow.load_play_data()
    assert ow.loaded_filenames == set(names)
    state, policy, z = ow.collect_all_loaded_data() if ow.dataset is None else ow.dataset
    # what the games contain: 8 symmetric rows per ply that has a row, each [(own, enemy), policy, z]
    from reversi_alpha_zero_amd.worker.self_play import rows_of_game
    want = []
    for plies, summ in games:
        want += rows_of_game([dict(p, saved_policy=p["saved_policy"]) for p in plies], summ["winner"])
    assert state.shape == (len(want), 2, 8, 8) and policy.shape == (len(want), 64) and z.shape == (len(want),)
    # row order inside the dataset follows the reference's dict of files; compare file by file instead
    at = 0
    for path in names:
        rows = read_game_data_from_file(path)
        s, p, zz = OptimizeWorker.convert_to_training_data(rows)
        for r, (ss, pp, z1) in zip(want[at:at + len(rows)], zip(s, p, zz)):
            (own, enemy), pol, zv = r
            sh = np.arange(64, dtype=np.uint64)
            assert np.array_equal(ss[0].reshape(64), ((np.uint64(own) >> sh) & np.uint64(1)).astype(ss.dtype))
            assert np.array_equal(ss[1].reshape(64), ((np.uint64(enemy) >> sh) & np.uint64(1)).astype(ss.dtype))
            assert np.array_equal(pp, np.asarray(pol, dtype=pp.dtype)) and z1 == zv
        at += len(rows)
    assert at == len(want)
    # ... and the drop-in's own loader agrees with the reference's on the same files
    from reversi_alpha_zero_amd.lib.data_helper import convert_to_training_data, read_game_data_from_file as my_read
    for path in names:
        a = OptimizeWorker.convert_to_training_data(read_game_data_from_file(path))
        b = convert_to_training_data(my_read(path))
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_reference_evaluate_worker_plays_a_game_between_drop_in_players(monkeypatch):
    import ref_harness as rh
    rh.install()
    from emu_util import EmuEngine
    import reversi_zero.worker.evaluate as ref_eval
    from reversi_alpha_zero_amd.agent import player as P

    golden = load_mcts_golden()
    blob = golden_net_blob(golden["net"])

    class Model:   # what the drop-in player needs of a model: its weights
        def to_blob(self):
            return blob

    made = []

    class EmuPlayer(P.ReversiPlayer):
        """The drop-in class with its engine built on the wave-emulator library (test rig: no GPU here)."""

        def _make_engine(self, info):
            from types import SimpleNamespace
            pc = P.effective_play_config(self.config, self.play_config)
            shim = SimpleNamespace(play=pc, play_data=self.config.play_data)
            eng = EmuEngine(shim, blob, 1, seed=info.seed, sims_hint=int(pc.simulation_num_per_move), record_root_w=True)
            made.append(eng)
            return eng

    monkeypatch.setattr(ref_eval, "ReversiPlayer", EmuPlayer)
    rcfg = rh.load_config("mini.yml", {"play": {"use_solver_turn": 0, "use_solver_turn_in_simulation": 0}})
    pc = rcfg.eval.play_config
    pc.simulation_num_per_move = 6
    pc.thinking_loop = 1
    pc.parallel_search_num = 2
    pc.use_solver_turn = 0
    pc.resign_threshold = None
    worker = ref_eval.EvaluateWorker(rcfg)
    random.seed(5)
    ng_win, best_is_black, (nb, nw) = worker.play_game(Model(), Model())
    assert ng_win in (None, 0, 1) and isinstance(best_is_black, bool)
    assert 0 < nb + nw <= 64 and len(made) == 2
    # the same game again is the same game (the players' random streams are keyed by (seed, game id), not by wall time)
    random.seed(5)
    again = worker.play_game(Model(), Model())
    assert again == (ng_win, best_is_black, (nb, nw))
    if ng_win is not None:
        black_won = nb > nw
        assert ng_win == (0 if black_won == best_is_black else 1)
