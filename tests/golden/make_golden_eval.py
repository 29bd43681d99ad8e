#!/usr/bin/env python
"""Generate tests/golden/eval_games.json: evaluation games played by the UNMODIFIED reference
(worker/evaluate.py:66-96 EvaluateWorker.play_game -> two agent/player.py ReversiPlayers with trees of their own ->
env/reversi_env.py -> lib/bitboard.py), best model against challenger, with raz-rng-v1 and two raznet-forward-v1 nets
injected.  Build container only:
    python tests/golden/make_golden_eval.py
Every number written is produced by reference code.  What is shimmed (module-level names, no reference file is edited):
    reversi_zero.worker.evaluate.ReversiPlayer   a subclass of the reference's own class that (a) receives the model's net through
                                                 the api= seam (agent/player.py:41) instead of a Keras model, (b) gets the py>=3.9
                                                 semaphore of oracle/ref_harness.py, (c) points the module-level random sources at
                                                 THIS player's stream for the duration of its action() - the reference's sources are
                                                 process-global, and a match gives every (model, game) pair a stream of its own
                                                 (reversi-alpha-zero_amd/worker/evaluate.py: seeds 2 s for the best model, 2 s + 1
                                                 for the challenger, event counters per game id)
    reversi_zero.worker.evaluate.random          random.Random(seed).random: the colour draw `random() < 0.5` of evaluate.py:71,
                                                 one draw per game in game order
    asyncio event loop                           ref_harness.VirtualTimeLoop: config.eval.play_config keeps PlayConfig's
                                                 parallel_search_num = 8 (config.py:101-110), so the players run on the exact-virtual-
                                                 time stage raz-sched-v1 is defined on (as tests/golden/mcts_par_games.json)
"""
import asyncio
import hashlib
import json
import os
import sys
from random import Random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402
import ref_selfplay as rs  # noqa: E402
from reversi_alpha_zero_amd.agent.model import ReversiNet  # noqa: E402

# name, config.eval.play_config overrides (over config.py:101-110: 400 sims, thinking_loop 1, c_puct 1, tau 0, no noise, resignation always
# enabled, and PlayConfig's defaults for the rest: parallel_search_num 8, use_solver_turn 50), config.play overrides (the GLOBAL section
# agent/player.py reads allowed_resign_turn :127, use_solver_turn_in_simulation :237, virtual_loss :264 from, whatever play_config
# says), sims, seed, first id, games
MATCHES = [
    # evaluate.py exactly as shipped (only the simulations per move reduced): root solver from turn 50 by play_config, solver inside
    # simulations from turn 50 by config.play
    ("eval_as_shipped", {}, {}, 24, 5, 0, 4),
    # no solver anywhere, one simulation in flight, an early resign threshold, resignation allowed from turn 12 by config.play
    ("eval_par1_resign_nosolver", {"parallel_search_num": 1, "resign_threshold": -0.05, "use_solver_turn": 0},
     {"use_solver_turn_in_simulation": 0, "allowed_resign_turn": 12}, 20, 6, 10, 3),
    # the two sections DISAGREE where the player reads the global one: play_config says virtual_loss 3 / in-simulation solver 60,
    # config.play says 2 / 48 (and the root solver from 52 is play_config's)
    ("eval_par4_sections_disagree", {"parallel_search_num": 4, "use_solver_turn": 52, "use_solver_turn_in_simulation": 60, "virtual_loss": 3,
                                     "resign_threshold": None}, {"use_solver_turn_in_simulation": 48, "virtual_loss": 2}, 16, 7, 0, 2),
]


def sparse(v):
    return {str(i): x for i, x in enumerate(v) if x != 0}


def play_match(config, blobs, seed, first, n):
    """n games of best (blobs[0]) against challenger (blobs[1]) through the reference's play_game."""
    rh.install()
    import reversi_zero.agent.player as rp
    import reversi_zero.worker.evaluate as rev
    from reversi_zero.env.reversi_env import Player
    models = {"best": object(), "ng": object()}
    apis = {id(models["best"]): rs.OracleNetAPI(blobs[0]), id(models["ng"]): rs.OracleNetAPI(blobs[1])}
    side_seed = {id(models["best"]): 2 * seed, id(models["ng"]): 2 * seed + 1}
    side_name = {id(models["best"]): "best", id(models["ng"]): "ng"}
    state = {"game_id": None, "plies": None}

    class Player_(rp.ReversiPlayer):
        def __init__(self, config_, model, play_config=None, enable_resign=True, mtcs_info=None, api=None):
            super().__init__(config_, None, play_config=play_config, enable_resign=enable_resign, mtcs_info=mtcs_info, api=apis[id(model)])
            self.sem = rh.CompatSemaphore(self.play_config.parallel_search_num)
            self._stream = rs.GameStream(side_seed[id(model)], state["game_id"])
            self._who = side_name[id(model)]

        def action_with_evaluation(self, own, enemy, callback_in_mtcs=None):
            saved = (rp.random, np.random.choice, np.random.dirichlet)
            rp.random, np.random.choice, np.random.dirichlet = self._stream.expand_uniform, self._stream.choice, self._stream.dirichlet
            try:
                res = super().action_with_evaluation(own, enemy, callback_in_mtcs=callback_in_mtcs)
            finally:
                rp.random, np.random.choice, np.random.dirichlet = saved
            key = rp.CounterKey(own, enemy, Player.black.value)
            solved = res.action is not None and float(res.n) == 999.0 and key not in self.var_n
            state["plies"].append({"who": self._who, "own": "0x%016x" % own, "enemy": "0x%016x" % enemy,
                                   "action": -1 if res.action is None else int(res.action), "n": float(res.n), "q": float(res.q),
                                   "root_n": None if solved else sparse([float(v) for v in self.var_n[key]])})
            return res

    saved_player, saved_random = rev.ReversiPlayer, rev.random
    old_loop = asyncio.get_event_loop_policy().get_event_loop()
    asyncio.set_event_loop(rh.VirtualTimeLoop())
    rev.ReversiPlayer, rev.random = Player_, Random(seed).random
    games = []
    try:
        worker = rev.EvaluateWorker(config)
        for g in range(n):
            state["game_id"], state["plies"] = first + g, []
            ng_win, best_is_black, score = worker.play_game(models["best"], models["ng"])
            games.append({"game_id": first + g, "ng_win": ng_win, "best_is_black": bool(best_is_black), "black_white": [int(score[0]), int(score[1])],
                          "plies": state["plies"]})
    finally:
        rev.ReversiPlayer, rev.random = saved_player, saved_random
        asyncio.get_event_loop().close()
        asyncio.set_event_loop(old_loop)
    return games


def main():
    nets = [ReversiNet(16, 1, 16).keras_init_(0).randomize_bn_(3), ReversiNet(16, 1, 16).keras_init_(21).randomize_bn_(22)]
    blobs = [n.to_blob() for n in nets]
    out = {"_generator": "tests/golden/make_golden_eval.py",
           "event_loop": "ref_harness.VirtualTimeLoop (exact virtual time, FIFO ties)",
           "nets": [{"role": role, "filters": 16, "res_layers": 1, "value_fc": 16, "keras_init_seed": s, "randomize_bn_seed": b,
                     "blob_sha256": hashlib.sha256(blob).hexdigest()} for role, s, b, blob in (("best", 0, 3, blobs[0]), ("challenger", 21, 22, blobs[1]))],
           "matches": []}
    keys = ["simulation_num_per_move", "thinking_loop", "required_visit_to_decide_action", "start_rethinking_turn", "c_puct", "noise_eps",
            "dirichlet_alpha", "change_tau_turn", "virtual_loss", "parallel_search_num", "resign_threshold", "allowed_resign_turn",
            "disable_resignation_rate", "use_solver_turn", "use_solver_turn_in_simulation", "share_mtcs_info_in_self_play"]
    for name, over, play_over, sims, seed, first, n in MATCHES:
        cfg = rh.load_config(None, {"eval": {"play_config": dict(over, simulation_num_per_move=sims)}, "play": play_over})
        games = play_match(cfg, blobs, seed, first, n)
        out["matches"].append({"name": name, "seed": seed, "first_game_id": first, "play_config_overrides": over, "config_play_overrides": play_over,
                               "resolved_play_config": {k: getattr(cfg.eval.play_config, k) for k in keys},
                               "resolved_config_play": {k: getattr(cfg.play, k) for k in ("allowed_resign_turn", "use_solver_turn_in_simulation", "virtual_loss",
                                                                                          "policy_decay_turn", "policy_decay_power")},
                               "games": games})
        for g in games:
            print(name, g["game_id"], "plies", len(g["plies"]), "ng_win", g["ng_win"], "best_is_black", g["best_is_black"], g["black_white"])
    path = os.path.join(HERE, "eval_games.json")
    with open(path, "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
