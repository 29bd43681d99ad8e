#!/usr/bin/env python
"""tools/ref_python_baseline.py — the CPU baseline SURVEY.md §8(d) names: the reference's OWN pure-Python self-play
(worker/self_play.py:139-175 SelfPlayWorker.start_game -> agent/player.py -> env -> lib/bitboard.py, unmodified source,
driven through oracle/ref_harness.py) on the host cores, with an in-process torch-CPU fp32 ReversiModelAPI
(torch.set_num_threads(1)) per worker process, P = os.cpu_count() processes.  Build container only (needs
/root/reference; the GPU box has none), so the result is COMMITTED as profiles/r2_cpu_baseline_reference_python.json
and carried by bench.py's JSON line as `cpu_baseline_reference_python` (cores stated).

    python tools/ref_python_baseline.py [--window 60]

Two workloads, the two bench.py reports: BASELINE configs[1] (mini net, 200 sims/move, mini.yml settings, one whole
game per process) and configs[2] (256x10 net, 800 sims/move, ch5.yml settings: a fixed wall-clock window, simulations
counted).  parallel_search_num at the yml value (throughput mode), thinking_loop = 1, solver off, like the GPU legs.
Reported, not optimised."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

NETS = {"mini": (16, 1, 16), "ch5": (256, 10, 256)}


class TorchCPUApi:
    """ReversiModelAPI.predict (agent/api.py:30-45) on the fp32 torch restatement of agent/model.py:28-72."""

    def __init__(self, net):
        self.net = net
        self.positions = 0

    def predict(self, x):
        import numpy as np
        import torch
        x = np.asarray(x)
        single = x.ndim == 3
        with torch.no_grad():
            p, v = self.net(torch.from_numpy(x.reshape(-1, 2, 8, 8).astype(np.float32)))
        self.positions += p.shape[0]
        p, v = p.numpy(), v.numpy()
        return (p[0], v[0]) if single else (p, v)


def worker(idx, which, window, q):
    import torch
    torch.set_num_threads(1)
    import ref_harness as rh
    import ref_selfplay as rs
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    rh.install()
    import reversi_zero.agent.player as rp
    yml, sims, par = ("mini.yml", 200, 4) if which == "mini" else ("ch5.yml", 800, 8)
    over = {"play": {"thinking_loop": 1, "use_solver_turn": 0, "use_solver_turn_in_simulation": 0,
                     "reset_mtcs_info_per_game": 1, "parallel_search_num": par}}
    cfg = rh.load_config(yml, over)
    net = ReversiNet(*NETS[which]).keras_init_(0).eval()
    api = TorchCPUApi(net)
    count = {"sims": 0}
    orig = rp.ReversiPlayer.start_search_my_move

    async def counted(self, own, enemy):
        count["sims"] += 1
        return await orig(self, own, enemy)
    rp.ReversiPlayer.start_search_my_move = counted
    t0 = time.perf_counter()
    games = 0

    class Stop(Exception):
        pass
    if window:   # fixed window: stop the game loop from inside the NN seam
        real_predict = api.predict

        def predict(x):
            if time.perf_counter() - t0 > window:
                raise Stop()
            return real_predict(x)
        api.predict = predict
    try:
        rs.run_reference_game(cfg, None, 0, idx, sims, api=api)
        games = 1
    except Stop:
        pass
    q.put({"sims": count["sims"], "games": games, "seconds": time.perf_counter() - t0, "nn_positions": api.positions})


def run(which, window, procs):
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(i, which, window, q)) for i in range(procs)]
    t0 = time.perf_counter()
    for p in ps:
        p.start()
    res = [q.get() for _ in ps]
    for p in ps:
        p.join()
    dt = time.perf_counter() - t0
    sims = sum(r["sims"] for r in res)
    busy = max(r["seconds"] for r in res)
    return {"value": sims / busy, "unit": "sims/s", "cores": procs, "kind": "reference-python",
            "games_per_hour": (sum(r["games"] for r in res) / busy * 3600.0) if not window else None,
            "sims": sims, "nn_positions": sum(r["nn_positions"] for r in res), "seconds": busy, "wall_seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=float, default=60.0, help="seconds of the 256x10 workload")
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    a = ap.parse_args()
    mp.set_start_method("spawn")
    out = {"what": "the reference's own pure-Python self-play (SelfPlayWorker.start_game, unmodified source via oracle/ref_harness.py), "
                   "torch-CPU fp32 net in process, torch.set_num_threads(1), one worker process per core",
           "where": f"build container, {os.cpu_count()} cores (the GPU box has no /root/reference): committed measurement, "
                    "tools/ref_python_baseline.py",
           "configs1_mini_200sims": dict(run("mini", 0.0, a.procs),
                                         sample=f"{a.procs} whole games (one per process), mini net, 200 sims/move, mini.yml settings, "
                                                "parallel_search_num 4, thinking_loop 1, solver off"),
           "configs2_ch5_800sims": dict(run("ch5", a.window, a.procs),
                                        sample=f"{a.window:.0f} s window per process, 256x10 net, 800 sims/move, ch5.yml settings, "
                                               "parallel_search_num 8, thinking_loop 1, solver off")}
    path = os.path.join(ROOT, "profiles", "r2_cpu_baseline_reference_python.json")
    with open(path, "wt") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
