"""Where a simulation step of the fused tree + net kernel (k_tree_net) spends its wave time: BASELINE configs[1] (4096 games, mini
net, 200 sims/move) on a library built with -DRAZ_FUSED_PROF (RAZ_LIB_PATH), phases timed by the shader clock (s_memtime, 100 MHz)
inside the kernel.  The ticks are WAVE time - four waves share a SIMD, so a phase's share is its share of a wave's life, waiting included.
Prints one JSON document."""
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
    games, sims = 4096, 200
    cfg = bench.mini_config(sims, 1)
    F, R, V = bench.NETS["mini"]
    net = DeviceNet(ReversiNet(F, R, V).keras_init_(0).to_blob(), dev)
    eng = SelfPlayEngine(cfg, net, n_games=games, seed=0, sims_hint=sims, fused=True, phase_profile=True)
    eng.start(0, sims)
    eng.step(50)
    eng.stats()
    eng.start(0, sims)
    torch.cuda.synchronize()
    p0 = eng.phase_profile()
    t0 = time.perf_counter()
    steps = 0
    while steps < 4000:
        eng.step(200)
        steps += 200
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.stats()
    p1 = eng.phase_profile()
    d = {k: p1[k] - p0[k] for k in p1}
    named = {"backup": d["backup"], "controller (begin_move / decide_move)": d["controller"], "select (descent)": d["select"],
             "forward in the wave": d["active_launches"]}
    total = sum(named.values())
    print(json.dumps({"steps": steps, "seconds": dt, "sims": st["total_sims"], "sims_per_s": st["total_sims"] / dt,
                      "wave_ticks_100MHz": named, "share": {k: v / total for k, v in named.items()},
                      "ticks_per_simulation": {k: v / st["total_sims"] for k, v in named.items()},
                      "inside": {"root_noise (in controller)": d["root_noise"], "node_load_wait (in select)": d["node_load_wait"],
                                 "expand_part_of_backup": d["expand_part_of_backup"], "first_arrival_probe": d["first_arrival_probe"]}}))


if __name__ == "__main__":
    main()
