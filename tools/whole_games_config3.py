#!/usr/bin/env python
"""tools/whole_games_config3.py — COMPLETE games of BASELINE configs[2]'s search (256x10 net on the split-f16 trunk, 800
sims/move, ch5.yml play settings, thinking_loop 1, solver off, parallel_search_num 1) at a reduced number of slots, with
continuous batching and pruned node pools (16 x sims nodes per game, as the headline bench sizes them):
    python tools/whole_games_config3.py [--slots 1024] [--games 1024] [--sims 800]
What it is for: bench.py times 20 steps of an 8192-game batch and extrapolates games/hour; this run measures, on whole
games, the two factors that extrapolation needs - simulations per net evaluation (simulations that end on finished
positions need none) and searched plies per game - and checks the first plies of sampled games against the oracle fed
with the device net's outputs.  One JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--check-plies", type=int, default=3)
    ap.add_argument("--leaf-cache-log2", type=int, default=0, help="cross-game evaluation cache of 2**k entries (0 = none)")
    ap.add_argument("--leaf-cache-max-discs", type=int, default=24)
    ap.add_argument("--progress", default=None, help="file to append a progress line to every ~2000 steps")
    a = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    import bench
    import oracle as O
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine, raw_from_packed
    dev = torch.device("cuda:0")
    cfg = bench.ch5_config(a.sims)
    blob = ReversiNet(256, 10, 256).keras_init_(0).to_blob()
    net = DeviceNet(blob, dev, kernel="f16x3")
    eng = SelfPlayEngine(cfg, net, n_games=a.slots, seed=0, sims_hint=a.sims, nodes_per_game=16 * a.sims, parts=1,
                         leaf_cache_log2=a.leaf_cache_log2 or None, leaf_cache_max_discs=a.leaf_cache_max_discs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = [0]

    def progress(steps, done, stc, *_):
        if a.progress and steps - last[0] >= 2048:
            last[0] = steps
            with open(a.progress, "at") as f:
                f.write(json.dumps({"seconds": time.perf_counter() - t0, "steps": steps, "games_done": done, "total_sims": stc["total_sims"],
                                    "nn_leaves": stc["nn_leaves"], "max_pool_used": stc["max_pool_used"]}) + "\n")
    outbox, st = eng.play_continuous(0, a.games, lambda gid: a.sims, chunk=256, on_chunk=progress)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    raw = raw_from_packed(*(outbox[k].cpu().numpy() for k in ("headers", "root_n", "summary")))
    searched = int((raw["headers"]["sims"] > 0).sum())
    out = {"workload": f"{a.games} complete self-play games on {a.slots} slots (continuous batching), 256x10 net (split-f16 trunk), {a.sims} sims/move, "
                       "ch5.yml play settings, thinking_loop=1, solver off, parallel_search_num=1, node pools 16 x sims (k_gc between harvests)",
           "seconds": dt, "steps": st["steps"], "ms_per_step": 1e3 * dt / st["steps"], "gc_runs": st["gc_runs"],
           "total_sims": st["total_sims"], "nn_leaves": st["nn_leaves"], "sims_per_net_evaluation": st["total_sims"] / st["nn_leaves"],
           "searched_plies_per_game": searched / a.games, "plies_per_game": float(raw["n_plies"].mean()),
           "sims_per_s_at_this_batch": st["total_sims"] / dt, "leaves_per_s_at_this_batch": st["nn_leaves"] / dt,
           "games_per_hour_at_this_batch": a.games / dt * 3600.0, "leaf_slot_occupancy": st["leaf_slot_occupancy"],
           "winners_black_white_draw": [int((raw["status"] & 0x0f == w).sum()) for w in (1, 2, 3)],
           "resigned_games": int(((raw["status"] & 0x20) != 0).sum()), "range_ok": net.range_ok(),
           "leaf_cache": dict(eng.leaf_cache_stats(), log2_entries=a.leaf_cache_log2, max_discs=a.leaf_cache_max_discs) if a.leaf_cache_log2 else None}
    if a.check_plies:
        nn = bench.device_nn(net)
        ocfg = O.play_cfg_from_config(cfg, parallel_search_num=1)
        checked = []
        for gid in (0, a.games - 1):
            plies, _ = O.selfplay_game(ocfg, None, 0, gid, a.sims, nn=nn, stop_after_plies=a.check_plies)
            for i, p in enumerate(plies):
                if int(raw["headers"][gid, i]["action"]) != p["action"] or [float(x) for x in raw["root_n"][gid, i]] != p["root_n"]:
                    raise AssertionError(f"game {gid} ply {i}: differs from the oracle")
            checked.append(gid)
        out["parity_check"] = {"result": "ok", "what": f"first {a.check_plies} plies (actions, root N after {a.sims} simulations each) of game ids {checked} == "
                                                      "the oracle searching with the device net's outputs"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
