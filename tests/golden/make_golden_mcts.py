#!/usr/bin/env python
"""Generate tests/golden/mcts_games.json: full self-play games played by the UNMODIFIED reference
(worker/self_play.py start_game -> agent/player.py -> env -> lib/bitboard.py) with raz-rng-v1 and
the raznet-forward-v1 net injected (oracle/ref_selfplay.py).  Build container only:
    python tests/golden/make_golden_mcts.py
Every number written is produced by reference code; this script only chooses configurations.
All variants use parallel_search_num=1 (the reference's only reproducible mode, SURVEY §7 hard
part 1) and the solver off (use_solver_turn=0; SURVEY §8(f) rank 1 is a later row).
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402
import ref_selfplay as rs  # noqa: E402
from reversi_alpha_zero_amd.agent.model import ReversiNet  # noqa: E402

NO_SOLVER = {"use_solver_turn": 0, "use_solver_turn_in_simulation": 0, "parallel_search_num": 1}

VARIANTS = [
    # name, yml, play overrides, play_data overrides, sims, (seed, game ids)
    ("agz", "alpha_go_zero.yml", {}, {}, 40, 11, [0, 1, 5]),
    ("agz_resign", "alpha_go_zero.yml", {"resign_threshold": -0.02, "allowed_resign_turn": 12,
                                         "disable_resignation_rate": 0.0}, {}, 25, 12, [0, 1, 2]),
    ("agz_noresign_draw", "alpha_go_zero.yml", {"resign_threshold": None}, {"drop_draw_game_rate": 0.5}, 20, 13, [4]),
    ("mini_shared", "mini.yml", {"reset_mtcs_info_per_game": 1}, {}, 20, 14, [0, 7]),
    ("default_shared_rethink", None, {"thinking_loop": 3, "required_visit_to_decide_action": 60,
                                      "start_rethinking_turn": 6}, {}, 25, 15, [2]),
    ("eval_like_nonoise", "alpha_go_zero.yml", {"noise_eps": 0, "change_tau_turn": 0}, {"save_policy_of_tau_1": True}, 30, 16, [0, 9]),
    ("ch5_tau4_cpuct5", "ch5.yml", {"thinking_loop": 1}, {}, 30, 17, [3]),
    # end-game solver ON (the reference's compiled Cython solver, lib/alt/reversi_solver_cython.pyx)
    ("mini_solver_as_shipped", "mini.yml", {"reset_mtcs_info_per_game": 1, "use_solver_turn": 50,
                                            "use_solver_turn_in_simulation": 50}, {}, 20, 18, [0, 3]),
    ("mini_solver_noresign", "mini.yml", {"reset_mtcs_info_per_game": 1, "use_solver_turn": 50,
                                          "use_solver_turn_in_simulation": 50, "resign_threshold": None}, {}, 16, 20, [2]),
    # BASELINE.json configs[0]: config/mini.yml self worker, 1 game, 100 sims/move, as shipped (shared tree,
    # thinking_loop 2, solver from turn 50) except parallel_search_num=1 (the reproducible mode)
    ("config0_mini_yml_100sims", "mini.yml", {"reset_mtcs_info_per_game": 1, "use_solver_turn": 50,
                                              "use_solver_turn_in_simulation": 50}, {}, 100, 0, [0]),
    # the sims/move of BASELINE.json's metric and configs[2..3] (800), ch5.yml play settings, one in flight
    ("ch5_800sims", "ch5.yml", {"thinking_loop": 1}, {}, 800, 38, [0]),
    ("agz_solver_52_50", "alpha_go_zero.yml", {"use_solver_turn": 52, "use_solver_turn_in_simulation": 50,
                                               "resign_threshold": None}, {}, 25, 19, [1]),
]


# parallel_search_num > 1: the unmodified reference player on the exact-virtual-time event loop
# (oracle/ref_harness.py VirtualTimeLoop = the stage raz-sched-v1 is defined on) -> mcts_par_games.json
PAR_VARIANTS = [
    ("mini_par4_as_shipped", "mini.yml", {"reset_mtcs_info_per_game": 1, "use_solver_turn": 50,
                                          "use_solver_turn_in_simulation": 50, "parallel_search_num": 4}, {}, 20, 31, [0, 1]),
    ("mini_par2_nosolver", "mini.yml", {"reset_mtcs_info_per_game": 1, "parallel_search_num": 2}, {}, 24, 35, [4]),
    ("default_par8_rethink", None, {"thinking_loop": 3, "required_visit_to_decide_action": 60, "start_rethinking_turn": 6,
                                    "parallel_search_num": 8}, {}, 30, 32, [0]),
    ("agz_par8_unshared", "alpha_go_zero.yml", {"parallel_search_num": 8}, {}, 40, 33, [2]),
    ("ch5_par8_cpuct5", "ch5.yml", {"thinking_loop": 1, "parallel_search_num": 8}, {}, 30, 34, [1]),
    ("agz_par16_resign", "alpha_go_zero.yml", {"resign_threshold": -0.02, "allowed_resign_turn": 12,
                                               "disable_resignation_rate": 0.0, "parallel_search_num": 16}, {}, 32, 36, [0]),
    # BASELINE.json configs[0] with nothing overridden but the tree reset: config/mini.yml, 1 game, 100 sims/move
    ("config0_mini_yml_100sims_as_shipped", "mini.yml", {"reset_mtcs_info_per_game": 1, "use_solver_turn": 50,
                                                         "use_solver_turn_in_simulation": 50, "parallel_search_num": 4}, {}, 100, 0, [0]),
    # ... and the same with ch5.yml's own (default) parallel_search_num 8
    ("ch5_800sims_par8", "ch5.yml", {"thinking_loop": 1, "parallel_search_num": 8}, {}, 800, 38, [0]),
    ("agz_par3_solver_52_50", "alpha_go_zero.yml", {"use_solver_turn": 52, "use_solver_turn_in_simulation": 50,
                                                    "resign_threshold": None, "parallel_search_num": 3}, {}, 25, 37, [1]),
]


# reset_mtcs_info_per_game > 1 (mini.yml ships 3): the worker's MCTSInfo is carried from game to game
# (worker/self_play.py:109-111,132-134); the game ids of a variant are played in order on ONE tree
# -> mcts_series_games.json.  mini.yml exactly as shipped (parallel_search_num 4, thinking_loop 2,
# solver from turn 50, reset every 3 games) and the same at parallel_search_num 1.
SERIES_VARIANTS = [
    ("mini_yml_as_shipped_3_games", "mini.yml", {}, {}, 16, 41, [0, 1, 2]),
    ("mini_yml_par1_3_games", "mini.yml", {"parallel_search_num": 1}, {}, 14, 42, [7, 8, 9]),
]


# dirichlet_alpha other than the shipped 0.5 (lib/bitboard.py:162-171 np.random.dirichlet([alpha]*k)): the general
# Gamma(alpha <= 1) sampler of raz-rng-v1 instead of the Box-Muller pairs -> mcts_alpha_games.json
ALPHA_VARIANTS = [
    ("agz_alpha03", "alpha_go_zero.yml", {"dirichlet_alpha": 0.3}, {}, 30, 51, [0, 3]),
    ("ch5_alpha1_shared", "ch5.yml", {"thinking_loop": 1, "dirichlet_alpha": 1.0, "noise_eps": 0.4}, {}, 30, 52, [1]),
    ("mini_alpha003_eps05", "mini.yml", {"reset_mtcs_info_per_game": 1, "dirichlet_alpha": 0.03, "noise_eps": 0.5}, {}, 20, 53, [2]),
]


def sparse(v):
    return {str(i): x for i, x in enumerate(v) if x != 0}


def main():
    import sys
    if "--alpha-only" in sys.argv:
        generate(ALPHA_VARIANTS, "mcts_alpha_games.json", virtual_time=False)
        return
    generate(ALPHA_VARIANTS, "mcts_alpha_games.json", virtual_time=False)
    generate(VARIANTS, "mcts_games.json", virtual_time=False)
    generate(PAR_VARIANTS, "mcts_par_games.json", virtual_time=True)
    generate(SERIES_VARIANTS, "mcts_series_games.json", virtual_time=True, series=True)


def generate(variants, fname, virtual_time, series=False):
    net = ReversiNet(16, 1, 16).keras_init_(0).randomize_bn_(3)
    blob = net.to_blob()
    out = {"_generator": "tests/golden/make_golden_mcts.py",
           "event_loop": "ref_harness.VirtualTimeLoop (exact virtual time, FIFO ties)" if virtual_time else "asyncio default (parallel_search_num=1: order-free)",
           "net": {"filters": 16, "res_layers": 1, "value_fc": 16, "keras_init_seed": 0, "randomize_bn_seed": 3,
                   "blob_sha256": hashlib.sha256(blob).hexdigest()},
           "games": []}
    for name, yml, play_over, pd_over, sims, seed, gids in variants:
        carry = {} if series else None   # one MCTSInfo for the variant's games, like one worker's `mtcs_info`
        for gi, gid in enumerate(gids):
            over = {"play": play_over if series else dict(NO_SOLVER, **play_over), "play_data": pd_over}   # variant keys win over NO_SOLVER
            cfg = rh.load_config(yml, over)
            ref = rs.run_reference_game(cfg, blob, seed, gid, sims, virtual_time=virtual_time, carry=carry)
            ref["series_index"] = gi if series else None
            rows = ref.pop("play_rows")
            plies = []
            for p in ref.pop("plies"):
                plies.append({"player": p["player"], "own": "0x%016x" % p["own"], "enemy": "0x%016x" % p["enemy"],
                              "action": p["action"], "n": p["n"], "q": p["q"], "has_row": p["has_row"],
                              "solved": bool(p.get("solved", False)),
                              "root_n": sparse(p["root_n"]), "root_w": sparse(p["root_w"]),
                              "saved_policy": sparse(p["saved_policy"]) if p["saved_policy"] else None})
            keys = ["thinking_loop", "required_visit_to_decide_action", "start_rethinking_turn", "c_puct",
                    "noise_eps", "dirichlet_alpha", "change_tau_turn", "virtual_loss", "parallel_search_num",
                    "resign_threshold", "allowed_resign_turn", "disable_resignation_rate", "use_solver_turn",
                    "use_solver_turn_in_simulation", "share_mtcs_info_in_self_play", "reset_mtcs_info_per_game"]
            ref["resolved_play"] = {k: getattr(cfg.play, k) for k in keys}
            ref["resolved_play_data"] = {"save_policy_of_tau_1": cfg.play_data.save_policy_of_tau_1,
                                         "drop_draw_game_rate": cfg.play_data.drop_draw_game_rate}
            g = dict(ref, variant=name, yml=yml, play_overrides=over["play"], play_data_overrides=pd_over,
                     plies=plies, black="0x%016x" % ref["black"], white="0x%016x" % ref["white"])
            g["play_rows_count"] = None if rows is None else len(rows)
            g["play_rows_sha256"] = None if rows is None else hashlib.sha256(json.dumps(rows).encode()).hexdigest()
            g["play_rows_head"] = None if rows is None else rows[:9]
            out["games"].append(g)
            print(name, gid, "plies", len(plies), "winner", ref["winner"], "resigned", ref["resigned_black"],
                  ref["resigned_white"], "rows", g["play_rows_count"], "nn", ref["nn_positions"])
    path = os.path.join(HERE, fname)
    with open(path, "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
