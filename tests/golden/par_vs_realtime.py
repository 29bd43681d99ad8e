#!/usr/bin/env python
"""tests/golden/par_vs_realtime.py — how far is raz-sched-v1 (the reference's asyncio loop in exact virtual
time) from what the reference does on the REAL event loop at parallel_search_num > 1?
The real interleaving depends on wall-clock timers and differs run to run, so only distributions can
be compared: for G self-play games per setting this prints, for the unmodified reference on the real
loop and on ref_harness.VirtualTimeLoop (== the oracle == the engine, bit for bit),
    leaves/sim  NN evaluations per simulation (collisions and finished positions lower it)
    top share   mean over plies of max(N) / sum(N) at the root (how concentrated the search is)
    entropy     mean over plies of the entropy of N / sum(N) (nats)
    plies       mean game length
Test infrastructure (it drives the oracle harness), build container only (needs /root/reference):
    python tests/golden/par_vs_realtime.py [--games 8] [--sims 30]"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def stats(games, sims):
    leaves = sum(g["nn_positions"] for g in games)
    searched = [p for g in games for p in g["plies"] if p["has_row"] and sum(p["root_n"]) > 1]
    nsims = sum(round(sum(p["root_n"])) for p in searched)   # (upper bound: root N accumulates across moves)
    top, ent = [], []
    for p in searched:
        n = [x for x in p["root_n"] if x > 0]
        s = sum(n)
        top.append(max(n) / s)
        ent.append(-sum(x / s * math.log(x / s) for x in n))
    sims_run = sum(sims * sum(1 for p in g["plies"] if p["has_row"]) for g in games)
    return {"leaves/sim": leaves / max(sims_run, 1), "top share": sum(top) / len(top), "entropy": sum(ent) / len(ent),
            "plies": sum(len(g["plies"]) for g in games) / len(games)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8)
    ap.add_argument("--sims", type=int, default=30)
    args = ap.parse_args()
    import ref_harness as rh
    import ref_selfplay as rs
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    blob = ReversiNet(16, 1, 16).keras_init_(0).randomize_bn_(3).to_blob()
    print(f"{args.games} games per row, {args.sims} sims/move, mini.yml play settings (solver off, tree reset per game)")
    print(f"{'parallel_search_num':>20} {'event loop':>12} " + " ".join(f"{k:>11}" for k in ("leaves/sim", "top share", "entropy", "plies")))
    for k in (1, 4, 8):
        over = {"play": {"parallel_search_num": k, "use_solver_turn": 0, "use_solver_turn_in_simulation": 0,
                         "reset_mtcs_info_per_game": 1, "thinking_loop": 1}}
        for loop, vt in (("real", False), ("virtual", True)):
            games = []
            for gid in range(args.games):
                cfg = rh.load_config("mini.yml", over)
                games.append(rs.run_reference_game(cfg, blob, 50 + k, gid, args.sims, virtual_time=vt))
            st = stats(games, args.sims)
            print(f"{k:>20} {loop:>12} " + " ".join(f"{st[m]:>11.4f}" for m in ("leaves/sim", "top share", "entropy", "plies")))


if __name__ == "__main__":
    main()
