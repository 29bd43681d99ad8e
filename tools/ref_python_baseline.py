#!/usr/bin/env python
"""tools/ref_python_baseline.py — the CPU baseline SURVEY.md §8(d) / BASELINE.json's north star name: the reference's OWN
pure-Python self-play (worker/self_play.py:139-175 SelfPlayWorker.start_game -> agent/player.py -> env -> lib/bitboard.py,
unmodified modules, driven through oracle/ref_harness.py) on the host cores, with an in-process torch-CPU fp32
ReversiModelAPI (torch.set_num_threads(1)) per worker process, one process per core, all processes released together and
stopped after the same wall-clock window; simulations (start_search_my_move invocations) are counted.

bench.py runs `measure()` on the BENCH BOX in the bench run (its `cpu_baseline`, kind "reference": the reference's own Python): there the
modules come from oracle/_ref (oracle/build_ref.py: the reference byte-compiled where it lies; /root/reference does not exist
on the GPU box).  Stand-alone:  python tools/ref_python_baseline.py [--window 20] [--procs N]

Workloads: "ch5" = BASELINE configs[2] (256x10 net, 800 sims/move, ch5.yml settings) and "mini" = configs[1] (mini net,
200 sims/move, mini.yml settings); parallel_search_num at the yml value (the reference's throughput mode), thinking_loop = 1,
solver off, like the GPU legs.  Reported, not optimised."""
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


def worker(idx, plan, q, barrier):
    """plan: [(which, window seconds)] run one after the other; every process waits at `barrier` before each window so that
    all cores are loaded during all of it."""
    try:
        import torch
        torch.set_num_threads(1)
        import ref_harness as rh
        import ref_selfplay as rs
        from reversi_alpha_zero_amd.agent.model import ReversiNet
        rh.install()
        import reversi_zero.agent.player as rp
        count = {"sims": 0}
        orig = rp.ReversiPlayer.start_search_my_move

        async def counted(self, own, enemy):   # COMPLETED simulations: the reference creates a move's coroutines all at once
            r = await orig(self, own, enemy)
            count["sims"] += 1
            return r
        rp.ReversiPlayer.start_search_my_move = counted

        class Stop(Exception):
            pass
        for which, window in plan:
            yml, sims, par = ("mini.yml", 200, 4) if which == "mini" else ("ch5.yml", 800, 8)
            over = {"play": {"thinking_loop": 1, "use_solver_turn": 0, "use_solver_turn_in_simulation": 0,
                             "reset_mtcs_info_per_game": 1, "parallel_search_num": par}}
            cfg = rh.load_config(yml, over)
            api = TorchCPUApi(ReversiNet(*NETS[which]).keras_init_(0).eval())
            real_predict = api.predict
            count["sims"] = 0
            games = 0
            barrier.wait(timeout=600)
            t0 = time.perf_counter()

            def predict(x):   # the window ends inside the NN seam (every simulation passes through it)
                if time.perf_counter() - t0 > window:
                    raise Stop()
                return real_predict(x)
            api.predict = predict
            try:
                while True:   # games idx, idx + P, ...: a process that finishes a game inside the window starts the next one
                    rs.run_reference_game(cfg, None, 0, idx + games * 4096, sims, api=api)
                    games += 1
            except Stop:
                pass
            q.put({"which": which, "sims": count["sims"], "games": games, "seconds": time.perf_counter() - t0, "nn_positions": api.positions})
    except BaseException as e:   # noqa: B902 - never leave the parent waiting
        q.put({"error": repr(e)})
        try:
            barrier.abort()
        except Exception:
            pass


def default_procs():
    """One process per host core, capped by memory (a worker holds torch + a 256x10 net: ~0.7 GB)."""
    n = os.cpu_count() or 1
    try:
        import psutil
        n = min(n, max(1, int(psutil.virtual_memory().available / (1 << 30) / 1.0)))
    except Exception:
        pass
    return n


def measure(plan=(("ch5", 20.0),), procs=None):
    """Run the plan on `procs` processes (default: one per core).  Returns {which: result dict}."""
    procs = procs or default_procs()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    barrier = ctx.Barrier(procs)
    t0 = time.perf_counter()
    ps = [ctx.Process(target=worker, args=(i, list(plan), q, barrier)) for i in range(procs)]
    for p in ps:
        p.start()
    res = []
    import queue as _queue
    deadline = time.perf_counter() + 600 + sum(w for _, w in plan)
    while len(res) < procs * len(plan):
        try:
            r = q.get(timeout=2.0)
        except _queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            if dead or time.perf_counter() > deadline:   # a child died before it could report (import error, OOM kill) / hung
                for p in ps:
                    p.terminate()
                raise RuntimeError(f"reference baseline: worker processes exited with {dead}" if dead else "reference baseline timed out")
            continue
        if "error" in r:
            for p in ps:
                p.terminate()
            raise RuntimeError("reference baseline worker failed: " + r["error"])
        res.append(r)
    for p in ps:
        p.join()
    wall = time.perf_counter() - t0
    import ref_harness as rh
    out = {}
    for which, window in plan:
        rs_ = [r for r in res if r["which"] == which]
        sims = sum(r["sims"] for r in rs_)
        busy = max(r["seconds"] for r in rs_)
        yml, n_sims, par = ("mini.yml", 200, 4) if which == "mini" else ("ch5.yml", 800, 8)
        out[which] = {"value": sims / busy, "unit": "sims/s", "cores": procs, "kind": "reference",
                      "sample": f"{window:.0f} s window on every one of {procs} processes (one per host core of THIS box, torch-CPU fp32 net in "
                                f"process, 1 thread each): the reference's SelfPlayWorker.start_game, {NETS[which][0]}x{NETS[which][1]} net, "
                                f"{n_sims} sims/move, {yml} settings, parallel_search_num {par}, thinking_loop 1, solver off; "
                                f"{sims} simulations, {sum(r['nn_positions'] for r in rs_)} net positions, "
                                f"{sum(r['games'] for r in rs_)} games finished inside the window"
                                + (f"; DEVIATION from BASELINE.md section 3, which names a 120 s window for the 256x10 net: {window:.0f} s here, so that the default "
                                   "bench run stays inside its time budget (sims/s of a steady loop does not depend on the window's length)" if which != "mini" and window < 120 else ""),
                      "sims": sims, "nn_positions": sum(r["nn_positions"] for r in rs_), "seconds": busy,
                      "reference_modules": rh.REFERENCE_ROOT, "host_cpu_count": os.cpu_count()}
    out["wall_seconds_incl_process_start"] = wall
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=float, default=20.0, help="seconds per workload")
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--workloads", default="ch5,mini")
    a = ap.parse_args()
    print(json.dumps(measure([(w, a.window) for w in a.workloads.split(",")], a.procs or None)))


if __name__ == "__main__":
    main()
