#!/usr/bin/env python
"""bench.py — MCTS self-play throughput of the batched engine on MI355X.  ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--games B] [--sims S] [--net mini|ch5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md §8(d) "Config 2"): B = 4096 concurrent games per GPU
from the initial position, mini.yml net (F=16, R=1, V=16; random-init, Keras initialisers, seed 0),
S = 200 simulations per move, play settings of config/mini.yml (c_puct 5, change_tau_turn 10,
Dirichlet root noise eps .25 / alpha .5, shared black/white tree, resign threshold -0.9 from turn
10) with the declared overrides thinking_loop = 1 and end-game solver off (use_solver_turn = 0).

A "step" is one pass of the hot path over the batch: the tree kernel (backup + per-move controller
+ PUCT descent for every live game) followed by ONE net evaluation of all gathered leaves.
Without --steps the timed region runs the whole batch of games to completion and K is the number
of steps that took; with --steps K exactly K steps are timed (from the opening position after W
warm-up steps on a throw-away start).  metric = MCTS simulations/sec (start_search_my_move
invocations / wall time, NN included, inputs resident in HBM), whole job over all GPUs.

At N = 1 the same JSON line also carries "mini_yml_parallel_search_num_4" (the same workload at mini.yml's own
parallel_search_num = 4, whole games) and "config2_8192x800_ch5": --config2-steps steps of BASELINE.json
configs[2] (8192 games, 256x10 net, 800 sims/move - the shape the metric's "800 sims/move" names),
whose dominant kernel is the MFMA convolution, with its own roofline object.  --config2-steps 0 skips both.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (≈6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
TREE_BYTES_PER_SELECTION = 142   # per selection 126 B read + 16 B backup RMW
TREE_BYTES_PER_SIM = 556         # expansion 280 B + leaf I/O 276 B

NETS = {"mini": (16, 1, 16), "ch5": (256, 10, 256)}


def bench_config(args):
    """Play settings: config/mini.yml:10-26 over the defaults of config.py:128-166, with the two
    declared overrides (thinking_loop=1, solver off)."""
    play = types.SimpleNamespace(
        simulation_num_per_move=args.sims, share_mtcs_info_in_self_play=bool(args.share),
        thinking_loop=1, required_visit_to_decide_action=40, start_rethinking_turn=10, c_puct=5,
        noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=10, virtual_loss=3, parallel_search_num=args.par,
        resign_threshold=-0.9, allowed_resign_turn=10, disable_resignation_rate=0.1,
        use_solver_turn=0, use_solver_turn_in_simulation=0)
    return types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))


def ch5_config(sims):
    """Play settings of config/ch5.yml:9-16 over config.py:128-166 (c_puct 5, change_tau_turn 4, shared
    tree, resign from turn 50), with the declared overrides of SURVEY.md §8(d) "Config 3":
    thinking_loop = 1 (ch5.yml:13 says 10), solver off, parallel_search_num = 1."""
    play = types.SimpleNamespace(
        simulation_num_per_move=sims, share_mtcs_info_in_self_play=True,
        thinking_loop=1, required_visit_to_decide_action=400, start_rethinking_turn=8, c_puct=5,
        noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=4, virtual_loss=3, parallel_search_num=1,
        resign_threshold=-0.9, allowed_resign_turn=50, disable_resignation_rate=0.1,
        use_solver_turn=0, use_solver_turn_in_simulation=0)
    return types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))


def mini_par_leg(dev, args, par):
    """The configs[1] workload at config/mini.yml's own parallel_search_num (4): that many simulations in
    flight per game on the deterministic raz-sched-v1 schedule (bit-exact vs the reference run on a
    virtual-time event loop, tests/golden/mcts_par_games.json).  Whole games, same settings otherwise."""
    import copy
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    a = copy.copy(args)
    a.par = par
    cfg = bench_config(a)
    F, R, V = NETS[args.net]
    net = DeviceNet(ReversiNet(F, R, V).keras_init_(0).to_blob(), dev)
    eng = SelfPlayEngine(cfg, net, n_games=args.games, seed=0, sims_hint=args.sims)
    eng.start(0, args.sims)
    eng.step(20)
    eng.stats()
    eng.start(0, args.sims)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0
    while True:
        eng.step(args.chunk)
        steps += args.chunk
        st = eng.stats()
        if st["max_pool_used"] + eng.nodes_per_step * args.chunk + 64 > eng.cfg.nodes_per_game:
            eng.gc(eng.cfg.nodes_per_game // 4)
        if st["finished_games"] >= args.games or steps > 80 * args.sims * 4:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": f"{args.games} concurrent self-play games/GPU, {args.net} net, {args.sims} sims/move, mini.yml play settings, "
                       f"thinking_loop=1, solver off, parallel_search_num={par} (mini.yml:19), whole games",
           "value": st["total_sims"] / dt, "unit": "sims/s", "games_per_hour": st["finished_games"] / dt * 3600.0,
           "steps": steps, "ms_per_step": 1e3 * dt / steps, "total_sims": st["total_sims"], "nn_leaves": st["nn_leaves"],
           "finished_games": st["finished_games"]}
    del eng, net
    torch.cuda.empty_cache()
    return out


def config2_leg(dev, steps, games=8192, sims=800):
    """BASELINE.json configs[2] (the shape the metric's "800 sims/move" is quoted on): 8192 concurrent
    games on one GPU, 256x10 net (ch5.yml has no model section => config.py:187-193), 800 sims/move.
    A whole batch of games is ~80 minutes at this size, so exactly `steps` steps are timed from the
    opening (every game has one leaf in every step there, so the rate is if anything pessimistic:
    later in the game terminal leaves cost no net evaluation).  Node pools are pruned by k_gc in
    real runs (16*S nodes per game = 147 GB for the batch)."""
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet, macs_per_position
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    F, R, V = NETS["ch5"]
    blob = ReversiNet(F, R, V).keras_init_(0).to_blob()
    cfg = ch5_config(sims)
    net = DeviceNet(blob, dev)
    # one slice: the net forward (~95 ms for 8192 positions) dwarfs the tree kernel, so there is nothing to
    # overlap, and the per-launch duration of the convolution is then its stand-alone duration
    parts = 1
    eng = SelfPlayEngine(cfg, net, n_games=games, seed=0, sims_hint=sims, nodes_per_game=16 * sims, parts=parts)
    eng.start(0, sims)
    eng.step(2)
    eng.stats()
    eng.start(0, sims)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tree_ms, net_ms = eng.step_timed(steps)   # HIP events around every launch, on the launching stream
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.stats()
    macs = macs_per_position(F, R, V)
    leaves_per_launch = st["nn_leaves"] / (steps * parts)
    net_avg_ms = net_ms / (steps * parts)
    ach = 2.0 * macs * leaves_per_launch / (net_avg_ms * 1e-3) / 1e12
    out = {"workload": f"{games} concurrent self-play games/GPU, 256x10 net (F{F} R{R} V{V}), {sims} sims/move, ch5.yml play "
                       f"settings, thinking_loop=1, solver off, first {steps} steps of the batch",
           "value": st["total_sims"] / dt, "unit": "sims/s", "leaves_per_s": st["nn_leaves"] / dt,
           "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "total_sims": st["total_sims"], "nn_leaves": st["nn_leaves"],
           "roofline": {"bound": "mfma", "kernel": "k_conv3x3_wide+k_conv0_wide+k_heads_wide (one net forward per slice)",
                        "algorithmic_flops_per_launch": 2.0 * macs * leaves_per_launch, "avg_kernel_ms": net_avg_ms,
                        "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS,
                        "traffic": None},
           "k_tree_avg_ms": tree_ms / (steps * parts),
           "sustained_net_TFLOPs": 2.0 * macs * st["nn_leaves"] / dt / 1e12,
           "bound_sims_per_s_at_f32_mfma_peak": FP32_PEAK_TFLOPS * 1e12 / (2.0 * macs)}
    del eng, net
    torch.cuda.empty_cache()
    return out


def cpu_baseline(cfg, blob, sims, budget_games):
    """The oracle (C port of agent/player.py + env + bitboard + the same net), one game per thread on
    the host cores (ctypes releases the GIL), same settings.  Reported, not optimised."""
    import concurrent.futures as cf
    import oracle as O
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=cfg.play.parallel_search_num)
    cores = min(os.cpu_count() or 1, budget_games)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(lambda gid: O.selfplay_game(ocfg, blob, 0, gid, sims)[1], range(cores)))
    dt = time.perf_counter() - t0
    total = sum(r["n_sims"] for r in res)
    return {"value": total / dt, "unit": "sims/s", "cores": cores, "kind": "port",
            "games_per_hour": cores / dt * 3600.0,
            "sample": f"{cores} complete self-play games, one per host thread, {sims} sims/move, same net and "
                      f"play settings (oracle/orc_mcts.c: C port of the reference player; {total} sims in {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = run the batch of games to completion")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--net", default="mini", choices=sorted(NETS))
    ap.add_argument("--share", type=int, default=1, help="share_mtcs_info_in_self_play (mini.yml: True)")
    ap.add_argument("--chunk", type=int, default=200, help="steps enqueued between completion polls")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-games", type=int, default=64)
    ap.add_argument("--nodes-per-game", type=int, default=0, help="node pool per game (0 = sized for whole games)")
    ap.add_argument("--parts", type=int, default=0, help="slices/streams the batch is stepped in (0 = engine default 3)")
    ap.add_argument("--graph", action="store_true", help="replay captured hipGraphs of 16 steps instead of launching kernel by kernel")
    ap.add_argument("--inner-max", type=int, default=0, help="max simulations completed per game per tree launch (0 = default 2)")
    ap.add_argument("--no-overlap", action="store_true", help="step the batch on one stream (no half-batch overlap)")
    ap.add_argument("--phase-profile", action="store_true", help="in-kernel s_memtime phase breakdown (perturbs timing)")
    ap.add_argument("--par", type=int, default=1,
                    help="play.parallel_search_num: simulations in flight per game (1 = the reference's reproducible mode, the headline; "
                         "mini.yml ships 4, the other configs 8: raz-sched-v1)")
    ap.add_argument("--config2-steps", type=int, default=12,
                    help="also time this many steps of BASELINE configs[2] (8192 games, 256x10 net, 800 sims/move) on rank 0 at N=1; 0 = skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path is device-only; there is no CPU fallback)")
    # RAZ_BENCH_SHARED_GPU=1 (test rig only): several ranks share the visible GPUs and rendezvous over
    # gloo, to exercise the N > 1 code path on a 1-GPU box; the reported line says so.
    shared_gpu = os.environ.get("RAZ_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if shared_gpu else dev   # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import __graft_entry__ as g
    if world > 1:   # one rank builds (a no-op when the shipped .so files are current), the others wait
        if local == 0:
            g.build()
        dist.barrier()
    g.build()
    from reversi_alpha_zero_amd.agent.model import ReversiNet, macs_per_position
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine

    F, R, V = NETS[args.net]
    blob = ReversiNet(F, R, V).keras_init_(0).to_blob()
    cfg = bench_config(args)
    net = DeviceNet(blob, dev)
    eng = SelfPlayEngine(cfg, net, n_games=args.games, seed=0, sims_hint=args.sims, phase_profile=args.phase_profile,
                         nodes_per_game=args.nodes_per_game or None,
                         single_stream=args.no_overlap, parts=args.parts, inner_max=args.inner_max,
                         use_graph=args.graph)
    first_id = rank * args.games

    # warm-up on a throw-away start (clocks, caches, code objects), then restart the same games
    eng.start(first_id, args.sims)
    eng.step(max(args.warmup, 1))
    eng.stats()
    eng.start(first_id, args.sims)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    steps, tree_ms, net_ms, timed_steps = 0, 0.0, 0.0, 0
    if args.steps > 0:
        left = args.steps
        while left > 0:
            n = min(args.chunk, left)
            if (steps // args.chunk) % 8 == 0:   # sample kernel durations with HIP events 1 chunk in 8
                a, b = eng.step_timed(n)
                tree_ms, net_ms, timed_steps = tree_ms + a, net_ms + b, timed_steps + n
            else:
                eng.step(n)
            steps, left = steps + n, left - n
        st = eng.stats()
    else:
        while True:
            if (steps // args.chunk) % 8 == 0:
                a, b = eng.step_timed(args.chunk)
                tree_ms, net_ms, timed_steps = tree_ms + a, net_ms + b, timed_steps + args.chunk
            else:
                eng.step(args.chunk)
            steps += args.chunk
            st = eng.stats()
            if st["max_pool_used"] + eng.nodes_per_step * args.chunk + 64 > eng.cfg.nodes_per_game:   # prune before a pool can overflow
                eng.gc(eng.cfg.nodes_per_game // 4)
            if st["finished_games"] >= args.games:
                break
            if steps > 80 * args.sims * 4 + 4000:
                raise SystemExit("engine did not finish")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([float(st["total_sims"]), float(st["finished_games"]), float(st["nn_leaves"]),
                        float(st["selections"]), elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[4].item())
    total_sims, finished, leaves, selections = (float(tot[i].item()) for i in range(4))

    # the single collective of the path: finished-game records -> rank 0 (timed separately)
    t1 = time.perf_counter()
    raw = eng.read_raw()
    gather_bytes = 0
    if world > 1:
        for k in ("headers", "root_n", "n_plies", "status", "resigned", "game_id", "final_black", "final_white"):
            t = torch.from_numpy(raw[k].view("u1").reshape(-1)).to(cdev)
            lst = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
            dist.gather(t, lst, dst=0)
            gather_bytes += t.numel() * world
        torch.cuda.synchronize()
    gather_s = time.perf_counter() - t1

    used_graph = eng.uses_graph()
    # standalone kernel durations (whole batch, one stream, nothing overlapping) on the opening phase,
    # for reference beside the in-run (overlapped) numbers
    standalone = None
    if rank == 0 and not args.no_overlap and args.games >= 256:
        eng.start(first_id, args.sims)
        eng.set_parts(1)
        eng.step(300)
        a, b = eng.step_timed(300)
        st1 = eng.stats()
        standalone = {"tree_ms": a / 300, "net_ms": b / 300, "steps_sampled": "300..600 of a fresh start",
                      "sims_per_step": st1["total_sims"] / 600.0, "leaves_per_step": st1["nn_leaves"] / 600.0,
                      "selections_per_step": st1["selections"] / 600.0}
        eng.set_parts(args.parts or 3)


    if rank == 0:
        macs = macs_per_position(F, R, V)
        lps = 1 if (args.no_overlap or args.games < 256) else (args.parts or 3)   # launches of each kernel per step
        per_launch_tree_bytes = (TREE_BYTES_PER_SELECTION * selections + TREE_BYTES_PER_SIM * total_sims) / world / max(steps * lps, 1)
        tree_avg_ms = tree_ms / max(timed_steps * lps, 1)
        net_avg_ms = net_ms / max(timed_steps * lps, 1)
        leaves_per_launch = leaves / world / max(steps * lps, 1)
        net_kernel = "k_net_mfma" if F in (16, 32, 64) else ("k_conv3x3_wide+heads" if F >= 128 and F % 64 == 0 else "k_net_wave")
        kern = {
            "k_tree": {"bound": "hbm", "avg_ms": tree_avg_ms, "algorithmic_bytes_per_launch": per_launch_tree_bytes,
                       "achieved": per_launch_tree_bytes / (tree_avg_ms * 1e-3) / 1e9 if tree_avg_ms else None,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s"},
            net_kernel: {"bound": "mfma", "avg_ms": net_avg_ms, "algorithmic_flops_per_launch": 2.0 * macs * leaves_per_launch,
                           "achieved": 2.0 * macs * leaves_per_launch / (net_avg_ms * 1e-3) / 1e12 if net_avg_ms else None,
                           "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s"},
        }
        for k in kern.values():
            k["frac"] = (k["achieved"] / k["peak"]) if k["achieved"] else None
        dom = "k_tree" if tree_avg_ms >= net_avg_ms else net_kernel
        # HBM bytes per launch from the committed PMC passes of this same command (separate rocprofv3 --pmc runs,
        # tools/run_profiles.sh -> tools/pmc_summary.py); only for the default workload they were collected on
        traffic, traffic_src = None, None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc", "final2_traffic.json")
        default_workload = (args.games, args.sims, args.net, args.share, lps, args.par) == (4096, 200, "mini", 1, 3, 1)
        if default_workload and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            for k in kern:
                t = tj["kernels"].get(k)
                if t:
                    kern[k]["traffic"] = t["hbm_bytes_per_launch"]
            traffic = kern[dom].get("traffic")
            traffic_src = "profiles/r1_pmc/final2_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; per slice launch)"
        roof = dict(kern[dom], kernel=dom, traffic=traffic, traffic_source=traffic_src)
        roof.pop("avg_ms")
        roof["avg_kernel_ms"] = kern[dom]["avg_ms"]
        out = {
            "metric": "MCTS simulations/sec (self-play, NN included)", "value": total_sims / elapsed, "unit": "sims/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 tree statistics + f32 net",
            "data": "synthetic (random-init net of the named architecture, games from the initial position)",
            "config": {"workload": f"{args.games} concurrent self-play games/GPU, {args.net} net (F{F} R{R} V{V}), "
                                   f"{args.sims} sims/move, mini.yml play settings, thinking_loop=1, solver off, parallel_search_num={args.par}"
                                   + ("" if args.steps == 0 else f", first {args.steps} steps only"),
                       "games_per_gpu": args.games, "sims_per_move": args.sims, "net": args.net,
                       "share_mtcs_info_in_self_play": bool(args.share), "whole_games": args.steps == 0,
                       "parallel_search_num": args.par,
                       "kernel_launches_per_step": lps * 2,
                       "overlap": f"{lps} slices on {lps} HIP streams" if lps > 1 else "single stream",
                       "launch": "hipGraph replay (16 steps per graph)" if used_graph else "kernel by kernel"},
            "sims_per_sec_per_gpu": total_sims / elapsed / world,
            "games_per_hour": finished / elapsed * 3600.0 if args.steps == 0 else None,
            "finished_games": finished, "total_sims": total_sims, "nn_leaves": leaves,
            "mean_selections_per_sim": selections / max(total_sims, 1.0),
            "roofline": roof, "kernels": kern,
            "record_gather": {"seconds": gather_s, "bytes": gather_bytes, "collective": ("gather (gloo test rig, ranks share GPUs)" if shared_gpu else "gather (RCCL)") if world > 1 else "none (1 GPU): D2H read"},
        }
        if standalone:
            sb = TREE_BYTES_PER_SELECTION * standalone["selections_per_step"] + TREE_BYTES_PER_SIM * standalone["sims_per_step"]
            sf = 2.0 * macs * standalone["leaves_per_step"]
            out["standalone_kernels"] = {
                "note": "one launch over the whole batch on one stream, nothing overlapping (opening phase)",
                "k_tree": {"avg_ms": standalone["tree_ms"], "achieved_GBps": sb / (standalone["tree_ms"] * 1e-3) / 1e9,
                           "frac_of_hbm_peak": sb / (standalone["tree_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                net_kernel: {"avg_ms": standalone["net_ms"], "achieved_TFLOPs": sf / (standalone["net_ms"] * 1e-3) / 1e12,
                             "frac_of_f32_mfma_peak": sf / (standalone["net_ms"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}}
            out["sustained_in_timed_region"] = {
                "net_TFLOPs": 2.0 * macs * leaves / world / elapsed / 1e12,
                "tree_algorithmic_GBps": (TREE_BYTES_PER_SELECTION * selections + TREE_BYTES_PER_SIM * total_sims) / world / elapsed / 1e9}
        if args.phase_profile:
            pp = eng.phase_profile()
            launches = max(pp["active_launches"], 1)
            out["phase_profile_ticks_per_active_game_launch"] = {k: v / launches for k, v in pp.items()}
        if world == 1 and args.config2_steps > 0 and (args.games, args.sims, args.net, args.par) == (4096, 200, "mini", 1):
            del eng, net   # the workspaces do not fit in HBM together
            torch.cuda.empty_cache()
            for key, leg in (("mini_yml_parallel_search_num_4", lambda: mini_par_leg(dev, args, 4)),
                             ("config2_8192x800_ch5", lambda: config2_leg(dev, args.config2_steps))):
                try:
                    out[key] = leg()
                except Exception as ex:   # never lose the main line over an extra leg
                    out[key] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, blob, args.sims, args.cpu_games)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
