#!/usr/bin/env python
"""How long does the device solver take?  The golden late-game positions (tests/golden/solver_kat.json) with >= 8 empties, exact mode
at the root, each armed on an engine slot of its own: wall time of the step that solves them, per position (one slot at a time, so
the time is that solve's) and all at once (the launch waits for the slowest).  One JSON line."""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    g.build()
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    with open(os.path.join(ROOT, "tests", "golden", "solver_kat.json")) as f:
        kat = json.load(f)
    play = types.SimpleNamespace(
        simulation_num_per_move=8, share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=40,
        start_rethinking_turn=10, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=10, virtual_loss=3,
        parallel_search_num=1, resign_threshold=None, allowed_resign_turn=10, disable_resignation_rate=0.0,
        use_solver_turn=46, use_solver_turn_in_simulation=46)
    cfg = types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
    cases = [(int(p["black"], 16), int(p["white"], 16), p["next_player"]) for p in kat["positions"]]
    cases = [c for c in cases if 8 <= 64 - bin(c[0] | c[1]).count("1") <= 14]
    blob = ReversiNet(16, 1, 16).keras_init_(0).to_blob()
    net = DeviceNet(blob, "cuda:0")
    per = {}
    for b, w, pl in cases:
        eng = SelfPlayEngine(cfg, net, n_games=1, seed=3, sims_hint=8)
        eng.start(0, 8)
        eng.set_position(0, b, w, pl, 8, enable_resign=False, one_move=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(1)
        torch.cuda.synchronize()
        per.setdefault(64 - bin(b | w).count("1"), []).append(1e3 * (time.perf_counter() - t0))
        del eng
    eng = SelfPlayEngine(cfg, net, n_games=len(cases), seed=3, sims_hint=8)
    eng.start(0, 8)
    for gi, (b, w, pl) in enumerate(cases):
        eng.set_position(gi, b, w, pl, 8, enable_resign=False, one_move=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(1)
    torch.cuda.synchronize()
    together = 1e3 * (time.perf_counter() - t0)
    print(json.dumps({"exact_root_solves_ms_by_empties": {str(e): {"n": len(v), "mean": sum(v) / len(v), "max": max(v)} for e, v in sorted(per.items())},
                      "all_%d_positions_in_one_launch_ms" % len(cases): together,
                      "cpu_oracle_c_port_for_comparison_ms": {"10 empties exact": {"mean": 44.7, "max": 193.1}, "9": {"mean": 9.7, "max": 19.0}, "8": {"mean": 1.7, "max": 4.1}}}))


if __name__ == "__main__":
    main()
