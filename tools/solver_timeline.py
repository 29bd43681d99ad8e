"""The mini.yml-as-shipped leg of BASELINE configs[1] (4096 games, 200 sims/move, thinking_loop 2, 4 in flight, end-game solver from turn
50; two-kernel pipeline, lock-step whole games) as a TIMELINE: wall time, simulations, finished games and the solver pool's counters per
chunk of simulation steps - where in a batch's life the steps get expensive and how many solves are in flight then.  One JSON document."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    dev = torch.device("cuda:0")
    games, sims, chunk = 4096, 200, int(os.environ.get("RAZ_TIMELINE_CHUNK", "100"))
    fused = os.environ.get("RAZ_TIMELINE_FUSED", "0") == "1"
    timed = os.environ.get("RAZ_TIMELINE_TIMED", "1") == "1"   # HIP events around the launches (tree + pool round | net), summed over the slices' streams
    cfg = bench.mini_config(sims, 4)
    cfg.play.thinking_loop, cfg.play.use_solver_turn, cfg.play.use_solver_turn_in_simulation = 2, 50, 50
    F, R, V = bench.NETS["mini"]
    net = DeviceNet(ReversiNet(F, R, V).keras_init_(0).to_blob(), dev)
    eng = SelfPlayEngine(cfg, net, n_games=games, seed=0, sims_hint=sims * 2, fused=fused)
    eng.start(0, sims)
    eng.step(50)
    eng.stats()
    eng.start(0, sims)
    torch.cuda.synchronize()
    rows, steps, t_all = [], 0, time.perf_counter()
    prev_s, prev_v = 0, None
    while True:
        t0 = time.perf_counter()
        tree_pool_ms, net_ms = eng.step_timed(chunk) if timed else (eng.step(chunk), (0.0, 0.0))[1]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps += chunk
        st = eng.stats()
        sv = eng.solver_stats()
        if eng.pool_nearly_full(st, chunk):
            eng.gc(min(eng.cfg.nodes_per_game // 4, st["max_pool_used"] // 2))
        d = {k: sv[k] - (prev_v[k] if prev_v else 0) for k in ("requests_posted", "answers", "wave_iterations", "busy_lane_iterations", "worker_wave_launches")}
        rows.append({"steps": steps, "ms_per_step": 1e3 * dt / chunk, "tree_and_pool_ms_per_step": tree_pool_ms / chunk, "net_ms_per_step": net_ms / chunk,
                     "ticks_per_wave_iteration_so_far": sv["ticks"]["ticks_per_wave_iteration"], "slow_phase_share_so_far": sv["ticks"]["slow_phase_share_of_worker_time"], "sims": st["total_sims"] - prev_s, "finished": st["finished_games"],
                     "idle_or_finished_slots": st.get("idle_or_finished_slots"), **d})
        prev_s, prev_v = st["total_sims"], sv
        if st["finished_games"] >= games or steps > 60000:
            break
    total = time.perf_counter() - t_all
    print(json.dumps({"fused": fused, "chunk": chunk, "steps": steps, "seconds": total, "sims": prev_s, "sims_per_s": prev_s / total,
                      "solver": {k: v for k, v in prev_v.items() if k != "ticks"}, "timeline": rows}))


if __name__ == "__main__":
    main()
