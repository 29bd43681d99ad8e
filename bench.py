#!/usr/bin/env python
"""bench.py — MCTS self-play throughput of the batched engine on MI355X.  ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` on its own re-executes itself through torch.distributed.run with N ranks.)

HEADLINE workload = BASELINE.json configs[2] / SURVEY.md §8(d) "Config 3", the configuration the metric's
"800 sims/move" is quoted on: 8192 concurrent games per GPU, the 256x10 residual net (config.py:187-193
defaults: ch5.yml has no model section; random-init, Keras initialisers, seed 0), S = 800 simulations per
move, play settings of config/ch5.yml:9-16 over config.py:128-166 (c_puct 5, change_tau_turn 4, shared
black/white tree, Dirichlet noise eps .25 / alpha .5, resign threshold -0.9 from turn 50), with the declared
overrides thinking_loop = 1 (ch5.yml:13 says 10), end-game solver off, parallel_search_num = 1 (the
reference's reproducible mode).  Rank r plays global game ids [r*8192, (r+1)*8192) (weak scaling).

A "step" is one pass of the hot path over the batch: the tree kernel (backup of the previous leaf, per-move
controller, PUCT descent to the next leaf, for every live game) followed by ONE net evaluation of all gathered
leaves.  A whole batch of games is ~50 000 steps at this size, so the timed region is K steps of the batch in
its STEADY STATE under continuous batching: slot i holds a game at a ply drawn from the time share each ply has
in a long run (calibrated in the same run, untimed; positions reached by on-device random playouts,
raz_step_batch), i.e. the mix of opening, middle-game and end-game positions (terminal leaves, deep boards) a
long run holds at any instant; --tree-warm + W warm-up steps are run on that state first, so the timed steps
search trees that already hold a few dozen simulations.  metric = MCTS simulations/sec = start_search_my_move invocations / wall time, NN included,
inputs resident in HBM, whole job over all GPUs.  games/hour is extrapolated from it (labelled).

ONE line of at most 8 KB is printed (compact_line: the contract keys, `roofline`, `cpu_baseline`, one short object per leg with
its value and its parity result); the full document with every leg's detail goes to bench_full.json (gpurun_out/ when present).
At N = 1 the run also holds: the parity spot checks (8 sampled games of the 8192-game batch from the opening AND 8 sampled slots
of the TIMED steady-state batch, root N / W bit for bit against the CPU oracle fed with the device net's outputs);
`whole_games_measured` = the headline configuration (all 8192 slots) PLAYED for a fixed 120 s window with continuous batching -
sims/s, games finished per hour, two games that finished inside the window == the oracle; `cpu_baseline` = the reference's own
pure-Python self-play timed in this run on this box's host cores (oracle/_ref, tools/ref_python_baseline.py); the headline on
the exact-f32 kernels; `ch5_yml_as_shipped` = the same batch with NO declared override (thinking_loop 10, parallel_search_num 8,
solver from turn 50) and the solver's measured share of a step; BASELINE configs[4] on one GPU; BASELINE configs[1] (4096 games x
mini net x 200 sims/move, whole games, spot-checked) on the fused tree + net kernel the worker runs for 16-filter nets, at
parallel_search_num 1 and mini.yml's 4, and on the two-kernel pipeline beside it; continuous batching; the bitboard-sweep HBM leg.
At N > 1 (and at N = 1 under RAZ_BENCH_NCCL_WORLD1=1, an RCCL group of one rank): the record gather is timed and its payload
verified (per-rank checksums), and a small whole-game batch is played sharded AND on rank 0 alone: the gathered records must be
byte-identical (SURVEY 8(d) Config 4's acceptance).
`config1_mini_yml_as_shipped*`: mini.yml's play section with no declared override (thinking_loop 2, 4 simulations in flight, solver
from turn 50) on both kernel forms - the solver-bound regime.

`worker_end_to_end_config1`: BatchedSelfPlayWorker.run() itself for 60 s (files written, RCCL group of one rank).  The default run
takes ~7 min on an MI355X box; --time-budget (600 s) keeps a slower box from delivering the line late: an extra leg whose nominal
duration no longer fits is not started and says so.

Environment (test rigs and profiling runs only; the driver's form uses none): RAZ_BENCH_SHARED_GPU=1 (N ranks on the visible GPUs over
gloo), RAZ_BENCH_NCCL_WORLD1=1 (the N > 1 path on an RCCL group of one rank), RAZ_BENCH_SOLVER_BUDGET=<iterations> (ch5_yml_as_shipped:
the per-launch solver budget), RAZ_BENCH_MINI_SHIPPED=1 (--net mini: the headline leg on mini.yml's play section as shipped).
"""
import argparse
import json
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (≈6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
F16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/F16 MFMA dense peak
TREE_BYTES_PER_SELECTION = 142   # per selection 126 B read + 16 B backup RMW
TREE_BYTES_PER_SIM = 556         # expansion 280 B + leaf I/O 276 B
MEAN_SEARCHED_PLIES = 58.6       # searched plies per self-play game (turn 0 is bypassed, agent/player.py:143-148);
                                 # replaced by the value measured in the whole-game leg when that leg runs

NETS = {"mini": (16, 1, 16), "ch5": (256, 10, 256)}


def mini_config(sims, par=1, share=True):
    """Play settings: config/mini.yml:10-26 over the defaults of config.py:128-166, with the two
    declared overrides (thinking_loop=1, solver off)."""
    play = types.SimpleNamespace(
        simulation_num_per_move=sims, share_mtcs_info_in_self_play=bool(share),
        thinking_loop=1, required_visit_to_decide_action=40, start_rethinking_turn=10, c_puct=5,
        noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=10, virtual_loss=3, parallel_search_num=par,
        resign_threshold=-0.9, allowed_resign_turn=10, disable_resignation_rate=0.1,
        use_solver_turn=0, use_solver_turn_in_simulation=0)
    return types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))


def ch5_config(sims, par=1):
    """Play settings of config/ch5.yml:9-16 over config.py:128-166 (c_puct 5, change_tau_turn 4, shared
    tree, resign from turn 50), with the declared overrides of SURVEY.md §8(d) "Config 3":
    thinking_loop = 1 (ch5.yml:13 says 10), solver off, parallel_search_num = 1."""
    play = types.SimpleNamespace(
        simulation_num_per_move=sims, share_mtcs_info_in_self_play=True,
        thinking_loop=1, required_visit_to_decide_action=400, start_rethinking_turn=8, c_puct=5,
        noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=4, virtual_loss=3, parallel_search_num=par,
        resign_threshold=-0.9, allowed_resign_turn=50, disable_resignation_rate=0.1,
        use_solver_turn=0, use_solver_turn_in_simulation=0)
    return types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))


# ------------------------------------------------------------------------------------------------------------
# parity spot checks (outside every timed region): the oracle is the CHECKER here, nothing measured runs on it
# ------------------------------------------------------------------------------------------------------------
def device_nn(dnet):
    """The reference's NN seam (ReversiPlayer(api=...), agent/player.py:41,346) served by the device net, one position
    per call: the oracle's search is then checked GIVEN the net's outputs (the net itself is checked against fp32
    torch in tests/).  One 16-byte upload, one raz_net_forward over a batch of 1 (the kernels are batch-invariant), one
    260-byte download per leaf, on the calling thread's current stream."""
    import ctypes
    import numpy as np
    import torch
    from reversi_alpha_zero_amd._native import lib, check
    dev = dnet.device
    boards = torch.zeros(2, dtype=torch.int64, device=dev)
    out = torch.zeros(65, dtype=torch.float32, device=dev)      # policy 64 | value
    host_in = torch.zeros(2, dtype=torch.int64).pin_memory()
    host_out = torch.zeros(65, dtype=torch.float32).pin_memory()
    need = lib.raz_net_scratch_bytes(dnet.filters, dnet.value_fc, 1)
    scratch = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    signed = lambda x: x - (1 << 64) if x >= 1 << 63 else x

    def nn(own, enemy):
        host_in[0], host_in[1] = signed(own), signed(enemy)
        with torch.cuda.device(dev):
            boards.copy_(host_in, non_blocking=True)
            s = torch.cuda.current_stream().cuda_stream
            check(lib.raz_net_forward(ctypes.byref(dnet.c), boards.data_ptr(), boards.data_ptr() + 8, None, out.data_ptr(), out.data_ptr() + 256,
                                      1, scratch.data_ptr() if need else None, need, s), "raz_net_forward")
            host_out.copy_(out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        a = host_out.numpy()
        return a[:64].astype(np.float32, copy=True), float(a[64])
    return nn


def spotcheck_partial_search(eng, cfg, dnet, seed, first_id, slots):
    """Games of the big batch in the middle of their first searched move: the engine's root statistics (N as u32, W as
    f64 bit patterns) of slot g after n completed simulations must equal the oracle's after an n-simulation search of
    the same game id (simulations are sequential at parallel_search_num 1, so the first n of 800 are an n-simulation
    search).  The oracle evaluates leaves through the device net (batch of 1): the same function as the batch's."""
    import numpy as np
    import oracle as O
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=1)
    nn = device_nn(dnet)
    raw = eng.read_raw()
    checked = []
    for g in slots:
        assert int(raw["n_plies"][g]) == 1 and int(raw["headers"][g, 0]["action"]) == 19, "turn-0 bypass (player.py:143-148)"
        black, white = int(raw["final_black"][g]), int(raw["final_white"][g])
        found, w, n, _ = eng.read_node(g, white, black, 1, 0)   # white to move: the player's own discs are "black" (player.py:95)
        if not found or int(n.sum()) == 0:
            raise AssertionError(f"spot check: slot {g} has no root statistics yet")
        # the first simulation of a move on a fresh tree expands the root and backs nothing up (player.py:283-327: "a leaf
        # expansion updates no N/W at the leaf"), every later one adds 1 to one root edge
        done = int(n.sum()) + 1
        plies, _ = O.selfplay_game(ocfg, None, seed, first_id + g, done, nn=nn, stop_after_plies=2)
        on, ow = np.array(plies[1]["root_n"]), np.array(plies[1]["root_w"])
        if not (np.array_equal(on, n.astype(np.float64)) and np.array_equal(ow.view(np.uint64), w.view(np.uint64))):
            bad = [int(i) for i in np.nonzero((on != n) | (ow.view(np.uint64) != w.view(np.uint64)))[0]]
            raise AssertionError(f"parity spot check FAILED: game id {first_id + g}, {done} simulations: root N/W differ from the oracle "
                                 f"at actions {bad}: engine N {[int(n[i]) for i in bad]} W {[float(w[i]) for i in bad]}, "
                                 f"oracle N {[on[i] for i in bad]} W {[ow[i] for i in bad]}")
        checked.append({"game_id": first_id + g, "sims": done})
    return checked


def spotcheck_steady_state(eng, cfg, dnet, seed, first_id, start, n_check=8):
    """The batch that was TIMED: slot g was put on the position start[g] = (black, white, player) (raz_engine_set_position,
    fresh tree, the game's random streams at their first events) and has since completed some simulations of its first
    move.  Its root statistics (N u32, W f64 bits) must equal the oracle's after the same number of simulations of game id
    first_id + g taken up at that position (orc_selfplay_game_from, pinned to the reference's worker in
    tests/test_oracle_mcts.py), the oracle evaluating leaves through the device net."""
    import numpy as np
    import oracle as O
    from reversi_alpha_zero_amd.engine import GAME_SUMMARY
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=1)
    nn = device_nn(dnet)
    b, w, p = start
    summ = eng.pack_records(0, eng.n_games, plies=1)["summary"].cpu().numpy().view(GAME_SUMMARY).reshape(-1)
    discs = np.unpackbits((b.view(np.uint64) | w.view(np.uint64)).view(np.uint8)).reshape(len(b), 64).sum(1)
    cand = np.nonzero((summ["n_plies"] == 0) & (discs > 4))[0]   # still in the first move; turn 0 is bypassed (player.py:143-148)
    if len(cand) < n_check:
        raise AssertionError("steady-state spot check: too few slots are still in their first move")
    order = np.argsort(discs[cand], kind="stable")   # spread the sample over the plies of the batch
    slots = [int(cand[order[i]]) for i in np.linspace(0, len(cand) - 1, n_check).astype(int)]
    checked = []
    for g in slots:
        own, enemy = (int(b[g]), int(w[g])) if int(p[g]) == 1 else (int(w[g]), int(b[g]))
        own, enemy = own & (2**64 - 1), enemy & (2**64 - 1)
        found, w64, n64, _ = eng.read_node(g, own, enemy, 1, 0)
        if not found or int(n64.sum()) == 0:
            raise AssertionError(f"steady-state spot check: slot {g} has no root statistics")
        done = int(n64.sum()) + 1   # the first simulation on a fresh tree expands the root and backs nothing up
        plies, _ = O.selfplay_game(ocfg, None, seed, first_id + g, done, nn=nn, stop_after_plies=1,
                                   start=(int(b[g]) & (2**64 - 1), int(w[g]) & (2**64 - 1), int(p[g])))
        on, ow = np.array(plies[0]["root_n"]), np.array(plies[0]["root_w"])
        if not (np.array_equal(on, n64.astype(np.float64)) and np.array_equal(ow.view(np.uint64), w64.view(np.uint64))):
            raise AssertionError(f"parity spot check FAILED: timed steady-state batch, slot {g} (game id {first_id + g}, ply {int(discs[g]) - 4}, "
                                 f"{done} simulations): root N/W differ from the oracle")
        checked.append({"game_id": first_id + g, "ply": int(discs[g]) - 4, "sims": done})
    return checked


def spotcheck_first_moves(eng, cfg, dnet, seed, first_id, start, ply, sims, plies=(47, 48, 49), want=4, max_extra_steps=384):
    """ch5.yml AS SHIPPED, the batch that was timed: slots taken up 1-3 plies before use_solver_turn play on (untimed) until they have
    decided their FIRST move - a search with 8 simulations in flight whose descents end in win/loss solves (suspended and resumed on the
    solver budget), re-thinking loops included - and that move (action, root visit counts) must equal the first move of the game the
    oracle plays for the same id from the same position with the same settings, leaves through the device net.  Returns what was checked."""
    import numpy as np
    import oracle as O
    from reversi_alpha_zero_amd.engine import GAME_SUMMARY, raw_from_packed
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=int(cfg.play.parallel_search_num))
    nn = device_nn(dnet)
    b, w, p = start
    cand = np.nonzero(np.isin(ply, plies))[0]
    if len(cand) < want:   # (not a parity failure: say so instead of losing the line)
        return {"result": "not_checked", "why": "too few slots of the batch start 1-3 plies before use_solver_turn"}
    cand = cand[np.linspace(0, len(cand) - 1, min(len(cand), 6 * want)).astype(int)]
    extra = 0
    import torch
    torch.cuda.synchronize()
    t0, sims0 = time.perf_counter(), eng.stats()["total_sims"]
    while True:
        summ = eng.pack_records(0, eng.n_games, plies=1)["summary"].cpu().numpy().view(GAME_SUMMARY).reshape(-1)
        ready = [int(g) for g in cand if summ["n_plies"][g] >= 1]
        if len(ready) >= want or extra >= max_extra_steps:
            break
        eng.step(16)
        extra += 16
        st = eng.stats()
        if eng.pool_nearly_full(st, 16):
            eng.gc(min(int(eng.cfg.nodes_per_game) // 4, st["max_pool_used"] // 2))
    torch.cuda.synchronize()
    sustained = {"steps": extra, "seconds": time.perf_counter() - t0, "sims_per_s": (eng.stats()["total_sims"] - sims0) / max(time.perf_counter() - t0, 1e-9),
                 "note": "the batch played on after the timed steps (no refill: finished games leave their slots idle), host clock incl. the polls: "
                         "what the as-shipped rate is once the games that reach the solver have queued up behind their budgets"} if extra else None
    if not ready:
        return {"result": "not_checked", "why": f"none of the sampled slots decided a move within {extra} extra steps", "played_on": sustained}
    checked = []
    for g in ready[:want]:
        pk = eng.pack_records(g, 1, plies=1)
        raw = raw_from_packed(*(pk[k].cpu().numpy() for k in ("headers", "root_n", "summary")))
        act, rn = int(raw["headers"][0, 0]["action"]), raw["root_n"][0, 0].astype(np.float64)
        oplies, _ = O.selfplay_game(ocfg, None, seed, first_id + g, sims, nn=nn, stop_after_plies=1,
                                    start=(int(b[g]) & (2**64 - 1), int(w[g]) & (2**64 - 1), int(p[g])))
        if act != oplies[0]["action"] or not np.array_equal(rn, np.array(oplies[0]["root_n"])):
            raise AssertionError(f"parity spot check FAILED: ch5.yml as shipped, slot {g} (game id {first_id + g}, taken up at ply {int(ply[g])}): "
                                 f"first move {act} / root N differ from the oracle's {oplies[0]['action']}")
        checked.append({"game_id": first_id + g, "taken_up_at_ply": int(ply[g]), "action": act, "root_visits": int(rn.sum()),
                        "solved_by_the_root_solver": bool(oplies[0].get("solved", 0))})
    return {"result": "ok", "what": "slots of the TIMED as-shipped batch taken up 1-3 plies before use_solver_turn: their first decided move (action, root N; 8 simulations "
                                    "in flight, in-simulation solves on the solver budget, re-thinking) == the oracle's for the same id from the same position, leaves "
                                    "through the device net", "untimed_steps_played_on": extra, "played_on": sustained, "games": checked}


def spotcheck_whole_games(eng, cfg, blob, seed, first_id, slots, sims):
    """Finished games of the batch against complete oracle games (C net): every action and every root visit count."""
    import concurrent.futures as cf
    import oracle as O
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=cfg.play.parallel_search_num)
    raw = eng.read_raw()
    with cf.ThreadPoolExecutor(max_workers=min(len(slots), os.cpu_count() or 1)) as ex:
        games = list(ex.map(lambda g: O.selfplay_game(ocfg, blob, seed, first_id + g, sims), slots))
    for g, (plies, summ) in zip(slots, games):
        n = int(raw["n_plies"][g])
        acts = [int(a) for a in raw["headers"][g, :n]["action"]]
        if acts != [p["action"] for p in plies] or (int(raw["status"][g]) & 0x0f) != summ["winner"]:
            raise AssertionError(f"parity spot check FAILED: game id {first_id + g}: moves differ from the oracle")
        for i, p in enumerate(plies):
            if [float(x) for x in raw["root_n"][g, i]] != p["root_n"]:
                raise AssertionError(f"parity spot check FAILED: game id {first_id + g}, ply {i}: root N differs")
    return [{"game_id": first_id + g, "plies": int(raw["n_plies"][g])} for g in slots]


# ------------------------------------------------------------------------------------------------------------
# headline: Config 3
# ------------------------------------------------------------------------------------------------------------
def stagger(eng, n, sims, seed, dev, ply_weights=None):
    """Continuous-batching steady state: slot i continues a game from a position at ply p_i in 0..58 reached by random
    playouts on the device (the sweep kernels; SURVEY §8(d) "value distributions").  p_i is drawn uniformly, or with
    `ply_weights` (the time share of each ply in a long run, see calibrate_ply_weights).  Returns the plies."""
    import numpy as np
    from bench_sweep import harvest_positions
    black, white, player, _ = harvest_positions(n, seed, dev, ply_weights)
    b, w, p = black.cpu().numpy(), white.cpu().numpy(), player.cpu().numpy()
    eng.set_positions(0, black.contiguous(), white.contiguous(), player.contiguous(), sims, enable_resign=True, one_move=False)
    occ = np.unpackbits((b.view(np.uint64) | w.view(np.uint64)).view(np.uint8)).reshape(n, 64).sum(1)
    eng._staggered = (b, w, p)   # where every slot was put (the steady-state spot check starts the oracle there)
    return occ.astype(np.int64) - 4


def slot_sims(eng):
    import numpy as np
    from reversi_alpha_zero_amd.engine import GAME_SUMMARY
    pk = eng.pack_records(0, eng.n_games, plies=1)
    return pk["summary"].cpu().numpy().view(GAME_SUMMARY).reshape(-1)["sims"].astype(np.int64)


def calibrate_ply_weights(eng, n, sims, seed, dev, first_id, steps=12):
    """In a long run with continuous batching a slot spends at ply p a time proportional to the steps a move takes there,
    S / r(p) with r(p) = simulations completed per step at ply p (1 while every simulation needs the net; more near the end
    of a game, where simulations end on finished positions and several complete per step).  r(p) is measured here on a
    uniformly staggered batch (untimed); the timed batch is then drawn with weights 1 / r(p)."""
    import numpy as np
    eng.start(first_id, sims)
    ply = stagger(eng, n, sims, seed, dev)
    eng.step(24)
    s0 = slot_sims(eng)
    eng.step(steps)
    r_slot = (slot_sims(eng) - s0) / float(steps)
    r = np.ones(59)
    for p in range(59):
        m = ply == p
        if m.any():
            r[p] = max(float(r_slot[m].mean()), 0.5)
    return (1.0 / r), r


def kernel_sources_sha256():
    """sha256 over the kernel sources of libraz (csrc/*.hip, csrc/*.h, include/raz.h, in name order): what a committed counter pass
    records (tools/pmc_summary.py) and what this run compares it with - a traffic figure measured on other sources is not reported."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "reversi-alpha-zero_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(ROOT, "include", "raz.h")]:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


PMC_DIR = "r6_pmc"   # the counter passes of this round's final build (profiles/README.md)


def committed_traffic(name):
    """A committed counter-traffic file of profiles/<PMC_DIR>/ with its provenance checked: (json, source note) - or (None, why not)
    when the file is missing, is half a measurement, or was measured on kernel sources other than the ones this run was built from."""
    rel = f"profiles/{PMC_DIR}/{name}"
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, f"{rel}: no counter pass committed for this build"
    with open(path) as f:
        t = json.load(f)
    if not t.get("fetch_pass_present") or not t.get("write_pass_present"):   # never report half a measurement
        return None, f"{rel}: FETCH_SIZE or WRITE_SIZE pass missing"
    prov = t.get("provenance") or {}
    if prov.get("kernel_sources_sha256") != kernel_sources_sha256():
        return None, f"{rel}: measured {prov.get('measured_utc', 'at an unknown time')} on OTHER kernel sources than this build's - not reported"
    return t, (f"{rel}, measured {prov.get('measured_utc')} UTC on these kernel sources (sha256 {prov.get('kernel_sources_sha256', '')[:12]}): separate rocprofv3 --pmc "
               "FETCH_SIZE and WRITE_SIZE passes of this command (tools/run_profiles.sh), FETCH doubled as MI355X_MICROARCH.md prescribes")


def conv_traffic():
    """HBM bytes per net forward from the committed PMC passes of THIS build (separate rocprofv3 --pmc runs of this command:
    tools/run_profiles.sh -> tools/pmc_summary.py); (None, reason) otherwise."""
    t, note = committed_traffic("headline_config3_traffic.json")
    return (t.get("net_forward_hbm_bytes_per_launch") if t else None), note


def sweep_traffic():
    """HBM bytes per launch of k_step / k_legal_moves from the committed PMC passes of tools/bench_sweep.py
    (tools/run_profiles.sh sweep -> tools/pmc_summary.py)."""
    path = os.path.join(ROOT, "profiles", PMC_DIR, "sweep_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def net_check_timed_leaves(eng, net, module, blob, dev, n_f64=256):
    """the net's VALUES on the timed batch.  The parity spot checks feed the oracle with the device net's outputs, so
    they pin the search given the net, not the net.  Here every row of the leaf exchange that the LAST timed step's forward evaluated
    (include/raz.h raz_engine_device_ptr 10-14) - the answers the timed tree kernels consumed, produced by the timed kernel (the
    split-f16 trunk, raznet-forward-v2) - is evaluated again by the exact-f32 kernels (raznet-forward-v1: bitwise the CPU oracle's
    chains) on the same positions; a sample of n_f64 of them also by the fp32 torch graph and by the graph in f64 (tools/trained_net.py
    F64Graph).  Raises when v2 is further than the north star's 1e-5 from v1 on any policy entry or value.  Untimed."""
    import torch
    from reversi_alpha_zero_amd.engine import DeviceNet
    from trained_net import F64Graph, planes_of
    t0 = time.perf_counter()
    ex = eng.leaf_exchange()
    idx = ex["active"].nonzero()[:, 0]
    own, enemy = ex["own"][idx].contiguous(), ex["enemy"][idx].contiguous()
    p_run, v_run = ex["policy"][idx].clone(), ex["value"][idx].clone()
    exact = DeviceNet(blob, dev, kernel="f32")
    p1, v1 = exact.predict_bitboards(own, enemy)
    dp, dv = float((p_run - p1).abs().max()), float((v_run - v1).abs().max())
    pick = torch.linspace(0, idx.numel() - 1, min(n_f64, idx.numel()), device=dev).long()
    planes = planes_of(own[pick], enemy[pick])
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    with torch.no_grad():
        pt, vt = module.to(dev).eval()(planes)
    p64, v64 = F64Graph(module, dev)(planes)
    module.cpu()
    worst = lambda p, v: {"policy": float((p.double() - p64).abs().max()), "value": float((v.double() - v64).abs().max())}
    out = {"n": int(idx.numel()), "max_abs_dp": dp, "max_abs_dv": dv, "tolerance": 1e-5,
           "what": "every leaf the last TIMED step's forward evaluated: the answers in the leaf exchange (timed kernel: " + str(net.kernel_name) +
                   ") against the exact-f32 kernels (raznet-forward-v1) on the same positions, max |d policy| and max |d value|",
           "sample_vs_the_graph_in_f64": {"n": int(pick.numel()), "timed_kernel": worst(p_run[pick], v_run[pick]), "exact_f32_kernels": worst(p1[pick], v1[pick]),
                                          "torch_fp32_on_this_gpu": worst(pt, vt[:, 0])},
           "seconds": time.perf_counter() - t0}
    del exact
    if not (dp <= 1e-5 and dv <= 1e-5):
        raise AssertionError(f"net_check: the timed net kernel is {dp:.3g} (policy) / {dv:.3g} (value) from the exact-f32 kernels on the timed leaves (> 1e-5)")
    out["result"] = "ok"
    return out


def headline_leg(args, dev, rank, world, cdev, group=False):
    import numpy as np
    import torch
    import torch.distributed as dist
    from reversi_alpha_zero_amd.agent.model import ReversiNet, macs_per_position
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    F, R, V = NETS[args.net]
    module = ReversiNet(F, R, V).keras_init_(0)
    blob = module.to_blob()
    cfg = ch5_config(args.sims) if args.net == "ch5" else mini_config(args.sims)
    if args.net == "mini" and os.environ.get("RAZ_BENCH_MINI_SHIPPED") == "1":   # profiling runs of the solver-bound regime (tools/sessions/r4_s22.sh)
        cfg.play.parallel_search_num, cfg.play.thinking_loop, cfg.play.use_solver_turn, cfg.play.use_solver_turn_in_simulation = 4, 2, 50, 50
    net = DeviceNet(blob, dev, kernel=args.net_kernel)
    parts = args.parts or (1 if args.net == "ch5" else 3)   # a 256x10 forward dwarfs the tree kernel: nothing to overlap
    # the cross-game evaluation cache as the worker attaches it for wide nets (2^26 entries, positions of <= 24 discs): the
    # synthetic steady-state batch comes from independent random playouts, so only its slots in the first plies find
    # positions to share - the line reports what the cache did; self-play from the opening shares more
    # (tools/whole_games_config3.py)
    cache_log2 = None if (args.no_leaf_cache or args.net != "ch5") else 26
    eng = SelfPlayEngine(cfg, net, n_games=args.games, seed=0, sims_hint=args.sims,
                         nodes_per_game=args.nodes_per_game or 16 * args.sims, parts=parts, leaf_cache_log2=cache_log2, leaf_cache_max_discs=24,
                         fused=bool(args.fused and args.net == "mini"))
    first_id = rank * args.games
    out = {}

    # (1) untimed: the batch from the opening, 24 steps, then 8 sampled slots against the oracle
    spot = None
    if rank == 0 and not args.no_spotcheck:
        eng.start(first_id, args.sims)
        eng.step(24)
        eng.stats()
        slots = [int(x) for x in np.linspace(0, args.games - 1, 8).astype(int)]
        t0 = time.perf_counter()
        checked = spotcheck_partial_search(eng, cfg, net, 0, first_id, slots)
        spot = {"result": "ok", "what": f"{len(checked)} game ids sampled from the {args.games}-game batch after 24 steps from the opening: "
                                        "root N (u32) and W (f64 bits) == the CPU oracle searching the same ids with the device net's outputs",
                "games": checked, "seconds": time.perf_counter() - t0}

    # (2) the steady state of continuous batching, W warm-up steps, K timed steps
    weights = None
    if not args.opening:
        weights, r = calibrate_ply_weights(eng, args.games, args.sims, 777 + rank, dev, first_id)
    eng.start(first_id, args.sims)
    if not args.opening:
        ply = stagger(eng, args.games, args.sims, 12345 + rank, dev, weights)
        out["steady_state"] = {"ply_distribution": "time share of ply p in a long run ~ 1 / r(p), r = simulations completed per step at ply p, "
                                                   "calibrated on a uniformly staggered batch in this run (untimed)",
                               "sims_per_step_by_ply_sampled": {str(p): round(float(r[p]), 3) for p in (1, 10, 20, 30, 40, 45, 50, 54, 56, 58)},
                               "mean_ply_of_the_timed_batch": float(ply.mean())}
    eng.step(max(args.warmup, 1) + (0 if args.opening else args.tree_warm))
    st0 = eng.stats()
    c0 = eng.leaf_cache_stats()
    torch.cuda.synchronize()
    if group:
        dist.barrier()
    t0 = time.perf_counter()
    tree_ms, net_ms = eng.step_timed(args.steps)   # HIP events around every launch, on the launching stream
    torch.cuda.synchronize()
    if group:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    st = eng.stats()
    c1 = eng.leaf_cache_stats()
    net_check = None
    if rank == 0 and not args.no_spotcheck and "f16x3" in (net.kernel_name or "") and not args.fused:
        net_check = net_check_timed_leaves(eng, net, module, blob, dev)
    d = {k: float(st[k] - st0[k]) for k in ("total_sims", "nn_leaves", "selections")}
    served = float((c1["hits"] - c0["hits"]) + (c1["in_batch_duplicates"] - c0["in_batch_duplicates"]))
    tot = torch.tensor([d["total_sims"], d["nn_leaves"], d["selections"], elapsed, float(st["finished_games"])],
                       dtype=torch.float64, device=cdev)
    if group:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[3].item())
    total_sims, leaves, selections = (float(tot[i].item()) for i in range(3))

    # (3) the single collective of the path: records -> rank 0, from HBM, cut to the games' plies
    t1 = time.perf_counter()
    gather = {"collective": "none (1 GPU)", "bytes": 0}
    if group:
        import numpy as np
        from reversi_alpha_zero_amd.worker.self_play import gather_packed
        last = {}

        def packed(plies):
            last["pk"] = eng.pack_records(0, args.games, plies)
            return last["pk"]
        raw, moved = gather_packed(packed, rank, world)
        torch.cuda.synchronize()
        gather = {"collective": f"{dist.get_backend()} gather of packed records from HBM ({world} ranks)", "bytes": moved,
                  "games": (len(raw["n_plies"]) if raw is not None else None), "seconds": time.perf_counter() - t1}
        # payload check (outside the timed figure): every rank's byte sums of what it packed, against the sums of its slice of
        # what arrived on rank 0
        mine = [int(last["pk"][k].contiguous().view(torch.uint8).to(torch.int64).sum().item()) for k in ("headers", "root_n")]
        sums = [None] * world
        dist.all_gather_object(sums, mine)
        if rank == 0:
            n = args.games
            for r in range(world):
                got = [int(np.ascontiguousarray(raw[k][r * n:(r + 1) * n]).view(np.uint8).astype(np.int64).sum()) for k in ("headers", "root_n")]
                if got != sums[r]:
                    raise AssertionError(f"record gather: the bytes that arrived from rank {r} differ from what it packed ({got} != {sums[r]})")
            gather["payload_check"] = f"byte sums of headers and root N of all {world} shards == what each rank packed"
    gather.setdefault("seconds", time.perf_counter() - t1)

    if rank != 0:
        return None
    spot_timed = None
    if not args.no_spotcheck and not args.opening:
        t2 = time.perf_counter()
        checked = spotcheck_steady_state(eng, cfg, net, 0, first_id, eng._staggered)
        spot_timed = {"result": "ok", "what": f"{len(checked)} slots of the TIMED steady-state batch (sampled across its plies), after the timed steps: root N (u32) "
                                              "and W (f64 bits) == the CPU oracle taking game id first_id + slot up at the slot's position, same number of "
                                              "simulations, leaves evaluated by the device net",
                      "games": checked, "seconds": time.perf_counter() - t2}
    macs = macs_per_position(F, R, V)
    launches = args.steps * parts
    leaves_per_launch = (leaves / world - served) / launches   # rows the net evaluated on this rank (the cache serves the rest)
    net_avg_ms, tree_avg_ms = net_ms / launches, tree_ms / launches
    ach = 2.0 * macs * leaves_per_launch / (net_avg_ms * 1e-3) / 1e12 if net_avg_ms else 0.0   # (--fused: no separate net kernel to time)
    traffic, traffic_src = conv_traffic() if args.net == "ch5" else (None, None)
    if traffic is not None and args.games != 8192:   # (the committed counter passes are of the default command: a forward of 8192 games per GPU)
        traffic, traffic_src = None, f"{traffic_src} - NOT reported: that is a forward of 8192 games per GPU, this run has {args.games}"
    v2 = "f16x3" in (net.kernel_name or "")
    net_kernel = {"ch5": ("one net forward = k_conv0_split + 20 x k_conv3x3_f16x3 (implicit GEMM on the f16 matrix cores, split operands: "
                          "3 MFMA flops per algorithmic flop; >99% of the forward) + k_heads_split") if v2 else
                         "one net forward = k_conv0_wide + 20 x k_conv3x3_wide (implicit GEMM on the f32 matrix cores, >99% of it) + k_heads_wide",
                  "mini": "k_net_mfma"}[args.net]
    peak = F16_PEAK_TFLOPS if v2 else FP32_PEAK_TFLOPS
    value = total_sims / elapsed
    out.update({
        "metric": "MCTS simulations/sec (self-play, NN included)", "value": value, "unit": "sims/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 tree statistics + f32 net" + (f" ({net.kernel_name})" if getattr(net, "kernel_name", None) else ""),
        "data": "synthetic (random-init net of the named architecture; positions from on-device random playouts, uniformly random ply)",
        "config": {"workload": f"BASELINE configs[2]: {args.games} concurrent self-play games/GPU, {args.net} net (F{F} R{R} V{V}), "
                               f"{args.sims} sims/move, ch5.yml play settings, thinking_loop=1, solver off, parallel_search_num=1; "
                               + ("first steps from the opening" if args.opening else
                                  "steady state of continuous batching (slots at plies 0..58 drawn from the time share of each ply in a long run)"),
                   "games_per_gpu": args.games, "sims_per_move": args.sims, "net": args.net,
                   "kernel_launches_per_step": parts * (2 if args.net == "mini" else 23), "slices": parts},
        "sims_per_sec_per_gpu": value / world, "leaves_per_sec": leaves / elapsed,
        "games_per_hour": value / (args.sims * MEAN_SEARCHED_PLIES) * 3600.0,
        "games_per_hour_note": f"EXTRAPOLATED (the whole-game leg did not run): sims/s / ({args.sims} sims/move x {MEAN_SEARCHED_PLIES} searched plies/game)",
        "total_sims": total_sims, "nn_leaves": leaves, "leaf_slot_occupancy": leaves / world / (args.steps * args.games),
        "mean_selections_per_sim": selections / max(total_sims, 1.0),
        "roofline": {"bound": "mfma", "kernel_name": ("k_conv3x3_f16x3" if v2 else "k_conv3x3_wide") if args.net == "ch5" else "k_net_mfma",
                     "kernel": net_kernel, "algorithmic_flops_per_launch": 2.0 * macs * leaves_per_launch,
                     "power_limited": ("socket power telemetry of this kernel (round 5, kernel unchanged; amdsmi at 10 Hz, profiles/r5/conv_f16x3_power_telemetry_*.jsonl): 1380 W of a 1400 W cap "
                                       "with the PPT violation active in every sample and the gfx clock at 2.00 GHz on random operands; 1004 W and 2.40 GHz on zero "
                                       "operands (same instruction stream, 1.26x faster); 77 % matrix-pipe busy (profiles/r4_pmc/)") if v2 else None,
                     "avg_kernel_ms": net_avg_ms, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                     "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                     # activations are 4 B per element in both kernel families: per conv layer one read + one write of
                     # n x F x 64 elements, a skip read every other layer, + the stem's write and the heads' read
                     "algorithmic_hbm_bytes_per_launch": (leaves_per_launch * F * 256.0 * (2 * 2 * R + R + 2)) if args.net == "ch5" else None,
                     "peak_note": ("f16 MFMA dense peak - the pipe the trunk runs on; every algorithmic (f32-accurate) flop costs 3 f16 MFMA "
                                   "flops, so this fraction cannot exceed 1/3" if v2 else "f32 MFMA dense peak"),
                     "executed_mfma_tflops": ach * (3.0 if v2 else 1.0), "executed_mfma_frac_of_pipe_peak": ach * (3.0 if v2 else 1.0) / peak,
                     # SURVEY 8(d) names the f32 MFMA peak as the denominator of the conv batch: the same achieved figure over it
                     "frac_of_f32_mfma_peak": ach / FP32_PEAK_TFLOPS},
        "kernels": {"k_tree": {"bound": "hbm", "avg_ms": tree_avg_ms,
                               "algorithmic_bytes_per_launch": (TREE_BYTES_PER_SELECTION * selections + TREE_BYTES_PER_SIM * total_sims) / world / launches}},
        "node_pools": {"nodes_per_game": int(eng.cfg.nodes_per_game), "bytes_per_game": int(eng.pool_bytes), "total_bytes": int(eng.pool_bytes) * args.games,
                       "engine_workspace_bytes": int(eng.workspace_bytes),
                       "layout": "compact nodes: 40 B header + 20 B per legal move (csrc/raz_engine.h)"},
        "bound_sims_per_s_per_gpu_at_f32_mfma_peak": FP32_PEAK_TFLOPS * 1e12 / (2.0 * macs),
        "leaf_cache": ({"entries_log2": cache_log2, "max_discs": 24, "served_from_the_table_in_the_timed_region": served,
                        "note": "the steady-state batch comes from independent random playouts: only its slots in the first plies share positions; "
                                "self-play from the opening shares more (profiles/r2/whole_games_config3_1024slots_leaf_cache.json)"}
                       if cache_log2 else None),
        "record_gather": gather, "parity_spotcheck": spot if spot else "skipped",
        "parity_spotcheck_timed_batch": spot_timed if spot_timed else "skipped",
        "net_check": net_check if net_check else "skipped (the timed net kernel is the exact-f32 one)" if "f16x3" not in (net.kernel_name or "") else "skipped",
    })
    k = out["kernels"]["k_tree"]
    k["achieved"] = k["algorithmic_bytes_per_launch"] / (tree_avg_ms * 1e-3) / 1e9 if tree_avg_ms else None
    k["peak"], k["unit"] = HBM_PEAK_GBS, "GB/s"
    k["frac"] = k["achieved"] / HBM_PEAK_GBS if k["achieved"] else None
    k["note"] = "latency-bound (one wave per game, dependent round trips): the HBM fraction is nominal"
    # counter traffic of the tree kernel: profiles/r4_pmc/headline_ktree_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes of this
    # command; "games_per_launch" says how many games the profiled launches covered - traffic per game is what is compared)
    tname = ("headline" if args.net == "ch5" else "config1") + "_traffic.json"
    tj, tnote = committed_traffic(tname)
    tpath = os.path.join(ROOT, "profiles", PMC_DIR, tname)
    tgames = None
    if tj is None:
        k["traffic"], k["traffic_source"] = None, tnote
    else:   # counter traffic of the same command (tools/run_profiles.sh): FETCH_SIZE / WRITE_SIZE passes
        k["traffic_source"] = tnote
        kt, tgames = tj.get("kernels", {}).get("k_tree"), (tj.get("games_per_launch") or ((tj.get("kernels", {}).get("k_tree") or {}).get("grid_threads") or 0) // 64 or None)
        if kt and k["algorithmic_bytes_per_launch"]:
            scale = (args.games / parts / tgames) if tgames else 1.0   # the profiled launches covered tgames games each
            kt = dict(kt, fetch_bytes_raw=kt["fetch_bytes_raw"] * scale, write_bytes=kt["write_bytes"] * scale)
            raw = kt["fetch_bytes_raw"] + kt["write_bytes"]
            k["traffic"] = raw
            k["traffic_over_algorithmic"] = raw / k["algorithmic_bytes_per_launch"]
            k["traffic_detail"] = {"fetch_bytes_raw": kt["fetch_bytes_raw"], "write_bytes": kt["write_bytes"],
                                   "with_fetch_doubled": (2 * kt["fetch_bytes_raw"] + kt["write_bytes"]) / k["algorithmic_bytes_per_launch"],
                                   "note": "FETCH_SIZE is calibrated (x2) for wide 16 B/lane reads only (MI355X_MICROARCH.md); the tree kernel reads 4-8 B per "
                                           "lane, so the raw figure is the traffic and the doubled one an upper bound; measured on the profiled run's launches "
                                           "(same command), algorithmic bytes from this run's selections and simulations",
                                   "source": os.path.relpath(tpath, ROOT)}
    out["_ply_weights"] = weights
    del eng, net
    torch.cuda.empty_cache()
    return out, blob, cfg


def exact_f32_leg(args, dev, blob, cfg, weights, steps=8):
    """The headline workload on the exact-f32 kernels (raznet-forward-v1: every output one k-ordered fmaf chain on the f32
    matrix cores, bit-identical to the CPU oracle), same steady-state batch recipe, fewer steps (a step is ~96 ms)."""
    import torch
    from reversi_alpha_zero_amd.agent.model import macs_per_position
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    F, R, V = NETS[args.net]
    net = DeviceNet(blob, dev, kernel="f32")
    eng = SelfPlayEngine(cfg, net, n_games=args.games, seed=0, sims_hint=args.sims, nodes_per_game=args.nodes_per_game or 16 * args.sims, parts=1)
    eng.start(0, args.sims)
    if weights is not None:
        stagger(eng, args.games, args.sims, 12345, dev, weights)
    eng.step(12)
    st0 = eng.stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tree_ms, net_ms = eng.step_timed(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.stats()
    sims, leaves = st["total_sims"] - st0["total_sims"], st["nn_leaves"] - st0["nn_leaves"]
    ach = 2.0 * macs_per_position(F, R, V) * (leaves / steps) / (net_ms / steps * 1e-3) / 1e12
    out = {"workload": "the headline workload on the exact-f32 kernels (raz_net.reserved = 0: k_conv3x3_wide on v_mfma_f32_32x32x2_f32), "
                       f"same steady-state batch recipe, {steps} steps",
           "value": sims / dt, "unit": "sims/s", "leaves_per_sec": leaves / dt, "ms_per_step": 1e3 * dt / steps,
           "roofline": {"bound": "mfma", "avg_kernel_ms": net_ms / steps, "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / FP32_PEAK_TFLOPS}}
    del eng, net
    torch.cuda.empty_cache()
    return out



# ------------------------------------------------------------------------------------------------------------
# whole games on the headline settings (driver-timed: measured, not extrapolated)
# ------------------------------------------------------------------------------------------------------------
def steady_window_leg(args, dev, blob, cfg, weights, slots=8192, seconds=120.0, chunk=32):
    """The headline CONFIGURATION over a fixed wall-clock window instead of 20 steps: `slots` (8192) resident games on the headline
    settings (256x10 net on the split-f16 trunk, 800 sims/move, ch5.yml play settings, thinking_loop 1, solver off,
    parallel_search_num 1), started in the steady state of continuous batching (every slot at a position drawn from the time share
    of each ply, as the timed batch of the headline) and then PLAYED for `seconds`: games finish, their slots restart from the
    opening on the next game id (raz_engine_harvest), pools are pruned by k_gc, the evaluation cache is attached as the worker
    attaches it.  sims/s and games/hour are measured over the window (host clock around synchronised chunks).  Returns (result,
    late-start games that finished inside the window - the complete-game check plays them again on the oracle)."""
    import numpy as np
    import torch
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine, raw_from_packed
    net = DeviceNet(blob, dev, kernel=args.net_kernel)
    cache_log2 = None if args.no_leaf_cache else int(os.environ.get("RAZ_BENCH_CACHE_LOG2", "26"))
    cache_discs = int(os.environ.get("RAZ_BENCH_CACHE_DISCS", "24"))   # (A/B of the evaluation cache's admission rule; the worker's defaults: 26, 24)
    eng = SelfPlayEngine(cfg, net, n_games=slots, seed=0, sims_hint=args.sims, nodes_per_game=args.nodes_per_game or 16 * args.sims, parts=1,
                         leaf_cache_log2=cache_log2, leaf_cache_max_discs=cache_discs)
    spare = slots   # ids for the refills of the window (a slot restarts at most a few times in two minutes)
    eng.start(0, args.sims)
    ply = stagger(eng, slots, args.sims, 2024, dev, weights)
    start_pos = eng._staggered
    outbox = eng.new_outbox(0, slots + spare)
    eng.step(args.tree_warm)
    nxt, done, steps, cap, gc_runs = slots, 0, 0, int(eng.cfg.nodes_per_game), 0
    # the worker's file emission INSIDE the window (worker/self_play.py:139-217 writes play_*.json inside the timed game): every game the
    # harvest finds finished goes - records packed in HBM, copied out, resignation bookkeeping - to BatchedSelfPlayWorker's background
    # writer (native row emitter on host threads, files on tmpfs), while the slots play on
    import shutil
    import tempfile
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker, _BackgroundWriter
    wroot = tempfile.mkdtemp(prefix="raz_window_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    wcfg = Config()
    for k_, v_ in vars(cfg.play).items():
        setattr(wcfg.play, k_, v_)
    wcfg.play_data.update(dict(nb_game_in_file=8, enable_ggf_data=False, max_file_num=100000, drop_draw_game_rate=0.0,
                               save_policy_of_tau_1=bool(cfg.play_data.save_policy_of_tau_1)))
    wrc = wcfg.resource
    wrc.data_dir, wrc.play_data_dir, wrc.self_play_ggf_data_dir = wroot, os.path.join(wroot, "play"), os.path.join(wroot, "ggf")
    wrc.self_play_game_idx_file = os.path.join(wroot, ".self-play-game-idx")
    os.makedirs(wrc.play_data_dir, exist_ok=True)
    ww = BatchedSelfPlayWorker(wcfg, blob, games_in_flight=slots, seed=0, device=str(dev))
    bw = _BackgroundWriter(ww)
    emitted = torch.zeros(slots + spare, dtype=torch.bool, device=dev)
    written, local_idx = 0, 1
    st0 = eng.stats()
    c0 = eng.leaf_cache_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        eng.step(chunk)
        steps += chunk
        st = eng.stats()   # synchronises
        if eng.pool_nearly_full(st, chunk):
            eng.gc(threshold=min(cap // 4, st["max_pool_used"] // 2))
            gc_runs += 1
        k = min(slots, slots + spare - nxt)
        h, r, skipped, playing = eng.harvest(outbox, nxt, [args.sims] * k)
        nxt += r
        done += h
        if h:
            new = torch.nonzero(outbox["done"].bool() & ~emitted).flatten()
            if len(new):
                emitted[new] = True
                fresh = raw_from_packed(*(outbox[k2][new].cpu().numpy() for k2 in ("headers", "root_n", "summary")))
                ww.bookkeep_raw(fresh)
                bw.submit(fresh, local_idx, written + len(new))
                local_idx += len(new)
                written += len(new)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bw.close()
    dt_drained = time.perf_counter() - t0
    files = sorted(os.listdir(wrc.play_data_dir))
    emission = {"what": "every game that finished inside the window was written as play_*.json by the worker's background writer (native row emitter, files of 8 games "
                        "on tmpfs) while the window ran; games/hour including emission = games / (window + the writer's drain after it)",
                "games_written": written, "files": len(files), "bytes_written": int(getattr(ww, "bytes_written", 0)),
                "writer_busy_seconds": bw.busy_seconds, "writer_busy_share_of_the_window": bw.busy_seconds / dt,
                "drain_after_the_window_seconds": dt_drained - dt, "games_per_hour_including_emission": written / dt_drained * 3600.0}
    if files:   # the last file holds rows: 8 symmetric rows per recorded ply, [[own, enemy], policy x 64, z]
        rows = json.load(open(os.path.join(wrc.play_data_dir, files[-1])))
        emission["last_file_rows"] = len(rows)
        if not rows or len(rows) % 8 or len(rows[0]) != 3 or len(rows[0][1]) != 64:
            raise AssertionError("steady window: the worker's play_*.json does not have the reference's row format")
    shutil.rmtree(wroot, ignore_errors=True)
    st = eng.stats()
    c1 = eng.leaf_cache_stats()
    sims, leaves = st["total_sims"] - st0["total_sims"], st["nn_leaves"] - st0["nn_leaves"]
    served = (c1["hits"] - c0["hits"]) + (c1["in_batch_duplicates"] - c0["in_batch_duplicates"])
    # finished games: rows of the outbox that are done; the late-start ones (slot ids < slots) are partial games by construction
    done_rows = torch.nonzero(outbox["done"]).flatten().cpu().numpy()
    late = [int(g) for g in done_rows if g < slots and ply[g] >= 50][:2]
    recs = {}
    if late:
        rows = torch.tensor(late, device=dev)
        raw = raw_from_packed(*(outbox[k][rows].cpu().numpy() for k in ("headers", "root_n", "summary")))
        b, w, p = start_pos
        for i, g in enumerate(late):
            n = int(raw["n_plies"][i])
            recs[g] = (raw["headers"][i, :n].copy(), raw["root_n"][i, :n].copy(), int(raw["status"][i]),
                       (int(b[g]) & (2**64 - 1), int(w[g]) & (2**64 - 1), int(p[g])))
    out = {"workload": f"the headline configuration for a fixed window: {slots} slots, 256x10 net ({net.kernel_name}), {args.sims} sims/move, ch5.yml play settings, "
                       "thinking_loop=1, solver off, parallel_search_num=1; slots start at positions drawn from the time share of each ply (the steady state of "
                       f"continuous batching) and are PLAYED for {seconds:.0f} s: finished games restart from the opening on the next id, node pools 16 x sims "
                       "pruned by k_gc, evaluation cache attached",
           "measured": "in this run, host clock around synchronised chunks of 32 steps (not extrapolated)",
           "value": sims / dt, "unit": "sims/s", "seconds": dt, "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "net_evaluations_per_s": leaves / dt, "sims_per_net_evaluation": sims / max(1, leaves),
           "leaf_slot_occupancy": leaves / max(1, steps * slots), "games_finished_in_the_window": int(done), "slots_restarted_from_the_opening": int(nxt - slots),
           "games_per_hour": done / dt * 3600.0,
           "games_per_hour_note": "games whose last move fell inside the window / window length: in the steady state that is the completion rate of whole games",
           "emission": emission,
           "gc_runs": gc_runs, "range_ok": net.range_ok(),
           "leaf_cache": ({"entries_log2": cache_log2, "max_discs": cache_discs, "served_from_the_table": int(served),
                           "share_of_leaf_requests": served / max(1, leaves)} if cache_log2 else None),
           "pool_bytes": int(eng.workspace_bytes)}
    del eng, net, outbox
    torch.cuda.empty_cache()
    return out, recs


def start_complete_game_checks(dev, args, blob, cfg, recs, sims=None):
    """Games of a leg against the games the CPU oracle plays for the same ids, the oracle evaluating every leaf through a device net
    of its own (the reference's NN seam, batch of 1) on its own stream: one host thread per game, started here and joined by the
    caller after other legs.  recs[id] = (ply headers, root N, status[, (black, white, next_player) the game was taken up at])."""
    import threading
    import numpy as np
    import torch
    import oracle as O
    from reversi_alpha_zero_amd.engine import DeviceNet
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=int(cfg.play.parallel_search_num))
    results = {}

    def check(gid):
        try:
            t0 = time.perf_counter()
            hdr, rn, status = recs[gid][:3]
            start = recs[gid][3] if len(recs[gid]) > 3 else None
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                dnet = DeviceNet(blob, dev, kernel=args.net_kernel)
                plies, summ = O.selfplay_game(ocfg, None, 0, gid, sims or args.sims, nn=device_nn(dnet), start=start)
            ok = len(plies) == len(hdr) and (status & 0x0f) == summ["winner"] and \
                [int(a) for a in hdr["action"]] == [p["action"] for p in plies] and \
                all(np.array_equal(rn[i].astype(np.float64), np.array(p["root_n"])) for i, p in enumerate(plies))
            results[gid] = {"ok": bool(ok), "plies": len(plies), "leaf_evaluations": int(summ["n_expand"]), "seconds": time.perf_counter() - t0}
            if start is not None:
                results[gid]["taken_up_at_ply"] = bin(start[0] | start[1]).count("1") - 4
        except BaseException as e:   # noqa: B902 - reported by the joiner
            results[gid] = {"ok": False, "error": repr(e)}
    threads = [threading.Thread(target=check, args=(gid,), daemon=True) for gid in recs]
    for t in threads:
        t.start()
    return threads, results


def join_complete_game_checks(threads, results, what, timeout=900.0):
    for t in threads:
        t.join(timeout)
    if any(t.is_alive() for t in threads):
        raise AssertionError("complete-game parity check did not finish")
    bad = {g: r for g, r in results.items() if not r.get("ok")}
    if bad:
        raise AssertionError(f"parity check FAILED: games of the bench differ from the oracle: {bad}")
    return {"result": "ok", "what": what,
            "games": [dict(game_id=g, **{k: v for k, v in r.items() if k != "ok"}) for g, r in sorted(results.items())]}


def config4_acceptance(dev, rank, world, cdev):
    """SURVEY 8(d) Config 4's acceptance on the real collective, scaled down: every rank plays its shard of a small
    whole-game batch (ids [64 r, 64 r + 64), mini net, 20 sims/move) and the records are gathered on rank 0 (RCCL under
    nccl); rank 0 then plays ALL world x 64 ids alone and the two packed record sets must be byte-identical."""
    import numpy as np
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    from reversi_alpha_zero_amd.worker.self_play import gather_packed
    per, sims = 64, 20
    cfg = mini_config(sims, 1)
    blob = ReversiNet(*NETS["mini"]).keras_init_(0).to_blob()

    def play(first, n):
        eng = SelfPlayEngine(cfg, DeviceNet(blob, dev), n_games=n, seed=0, sims_hint=sims)
        eng.start(first, sims)
        eng.run(chunk=128)
        return eng
    eng = play(rank * per, per)
    raw, moved = gather_packed(lambda plies: eng.pack_records(0, per, plies), rank, world)
    if rank != 0:
        return None
    alone = play(0, per * world)
    pk = alone.pack_records(0, per * world, plies=raw["headers"].shape[1])
    from reversi_alpha_zero_amd.engine import raw_from_packed
    one = raw_from_packed(*(pk[k].cpu().numpy() for k in ("headers", "root_n", "summary")))
    same = all(np.array_equal(np.asarray(raw[k]).view(np.uint8), np.asarray(one[k]).view(np.uint8))
               for k in ("headers", "root_n", "n_plies", "status", "game_id", "final_black", "final_white", "resigned"))
    if not same:
        raise AssertionError("Config 4 acceptance FAILED: the gathered records of the sharded batch differ from rank 0 playing all ids alone")
    return {"result": "ok", "what": f"{per * world} complete games (mini net, {sims} sims/move): {world} ranks x {per} ids gathered on rank 0 == rank 0 playing all "
                                    "ids alone, byte for byte (headers, root N, summaries)", "gathered_bytes": moved}


def cpu_baseline_reference(windows=(("ch5", 20.0), ("mini", 10.0))):
    """`cpu_baseline`: the reference's own pure-Python self-play on THIS box's host cores, in this run
    (tools/ref_python_baseline.py; modules from /root/reference or, on the GPU box, oracle/_ref)."""
    import ref_harness
    if not ref_harness.reference_available():
        return None
    import ref_python_baseline
    r = ref_python_baseline.measure(list(windows))
    out = dict(r["ch5"])
    out["configs1_mini_200sims"] = r.get("mini")
    out["wall_seconds_incl_process_start"] = r["wall_seconds_incl_process_start"]
    return out


# ------------------------------------------------------------------------------------------------------------
# extra legs at N = 1
# ------------------------------------------------------------------------------------------------------------
def agz_config(sims):
    """Play settings of config/alpha_go_zero.yml:5-18 over config.py:128-166 (unshared trees, c_puct 5, change_tau_turn 10,
    resign from turn 20, solver off, thinking_loop 1; Dirichlet noise eps .25 / alpha .5), parallel_search_num = 1."""
    play = types.SimpleNamespace(
        simulation_num_per_move=sims, share_mtcs_info_in_self_play=False,
        thinking_loop=1, required_visit_to_decide_action=400, start_rethinking_turn=8, c_puct=5,
        noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=10, virtual_loss=3, parallel_search_num=1,
        resign_threshold=-0.9, allowed_resign_turn=20, disable_resignation_rate=0.1,
        use_solver_turn=0, use_solver_turn_in_simulation=0)
    return types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=False))


def config5_leg(args, dev, blob, weights, games=8192, sims=3200, steps=10):
    """BASELINE configs[4] / SURVEY 8(d) "Config 5" on ONE GPU's share of it: 8192 concurrent games, alpha_go_zero.yml play
    settings, the 256x10 net (the yml defines no deeper one), S = 3200 (the BASELINE's override), Dirichlet root noise on.
    Node pools of 16 x S compact nodes per game, as the worker sizes them.  Same steady-state batch recipe as the headline,
    fewer steps; the trees hold a few dozen simulations, so k_tree's share of a step is a lower bound of a long run's."""
    import torch
    from reversi_alpha_zero_amd.agent.model import macs_per_position
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    cfg = agz_config(sims)
    net = DeviceNet(blob, dev, kernel=args.net_kernel)
    eng = SelfPlayEngine(cfg, net, n_games=games, seed=0, sims_hint=sims, nodes_per_game=16 * sims, parts=1,
                         leaf_cache_log2=None if args.no_leaf_cache else 26, leaf_cache_max_discs=24)
    eng.start(0, sims)
    stagger(eng, games, sims, 4242, dev, weights)
    eng.step(40)
    st0 = eng.stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tree_ms, net_ms = eng.step_timed(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.stats()
    sims_done, leaves = st["total_sims"] - st0["total_sims"], st["nn_leaves"] - st0["nn_leaves"]
    out = {"workload": f"BASELINE configs[4] on one GPU: {games} concurrent self-play games, alpha_go_zero.yml play settings (unshared trees, c_puct 5, "
                       f"change_tau_turn 10), 256x10 net ({net.kernel_name}), {sims} sims/move, Dirichlet root noise on, thinking_loop=1, "
                       f"parallel_search_num=1; steady-state ply mix, {steps} timed steps after 40 warm-up steps",
           "value": sims_done / dt, "unit": "sims/s", "leaves_per_sec": leaves / dt, "ms_per_step": 1e3 * dt / steps,
           "k_tree_avg_ms": tree_ms / steps, "net_forward_avg_ms": net_ms / steps,
           "nodes_per_game": int(eng.cfg.nodes_per_game), "node_pool_bytes_per_game": int(eng.pool_bytes),
           "node_pools_total_bytes": int(eng.pool_bytes) * games, "engine_workspace_bytes": int(eng.workspace_bytes),
           "note": "round 2's 1408-byte nodes needed 590 GB for these pools; the compact nodes (40 B + 20 B per legal move) need "
                   f"{int(eng.pool_bytes) * games / 1e9:.0f} GB"}
    del eng, net
    torch.cuda.empty_cache()
    return out


def ch5_shipped_config(sims):
    """config/ch5.yml:9-16 over config.py:128-166 with NOTHING overridden but the simulations per move (BASELINE configs[2]: 800):
    thinking_loop 10 / required_visit_to_decide_action 400 / start_rethinking_turn 8 (config.py:133-135), parallel_search_num 8
    (config.py:142), use_solver_turn = use_solver_turn_in_simulation = 50 (config.py:154-155)."""
    cfg = ch5_config(sims, par=8)
    cfg.play.thinking_loop = 10
    cfg.play.use_solver_turn = 50
    cfg.play.use_solver_turn_in_simulation = 50
    return cfg


def ch5_as_shipped_leg(args, dev, blob, weights, games=8192, steps=6, warm=12):
    """The headline batch with ch5.yml AS SHIPPED (no declared override left: thinking_loop 10, parallel_search_num 8, the end-game
    solver from turn 50): 8 simulations in flight per game (k_tree_par: up to 8 x 8192 leaves per net batch), re-think loops, the
    exhaustive solver inside the tree kernel.  Same steady-state batch recipe as the headline.  Timed twice - with the solver and
    with use_solver_turn = 0 - so that the solver's share of a step (it runs to completion inside the launch, every wave of the
    batch waiting for the slowest DFS) is a measured number."""
    import torch
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    res = {}
    for label in ("as_shipped", "root_solver_only", "solver_off"):
        cfg = ch5_shipped_config(args.sims)
        if label == "solver_off":
            cfg.play.use_solver_turn = 0
        if label != "as_shipped":
            cfg.play.use_solver_turn_in_simulation = 0
        net = DeviceNet(blob, dev, kernel=args.net_kernel)
        # a move's search holds up to thinking_loop x sims simulations (two nodes each with mirror keys) before k_gc prunes what the
        # game has left behind: 5 x that, as the 16 x sims of the headline pools
        nodes = 5 * 10 * args.sims
        eng = SelfPlayEngine(cfg, net, n_games=games, seed=0, sims_hint=args.sims, nodes_per_game=nodes, parts=1,
                             leaf_cache_log2=None if args.no_leaf_cache else 26, leaf_cache_max_discs=24,
                             solver_budget=int(os.environ.get("RAZ_BENCH_SOLVER_BUDGET", "0")))
        eng.start(0, args.sims)
        ply = stagger(eng, games, args.sims, 31337, dev, weights)
        eng.step(warm)
        st0 = eng.stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tree_ms, net_ms = eng.step_timed(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = eng.stats()
        sims_done, leaves = st["total_sims"] - st0["total_sims"], st["nn_leaves"] - st0["nn_leaves"]
        res[label] = {"value": sims_done / dt, "unit": "sims/s", "ms_per_step": 1e3 * dt / steps, "k_tree_par_ms_per_step": tree_ms / steps,
                      "net_forward_ms_per_step": net_ms / steps, "net_evaluations_per_step": leaves / steps, "sims_per_step": sims_done / steps,
                      "leaf_slot_occupancy": leaves / (steps * games * 8), "engine_workspace_bytes": int(eng.workspace_bytes)}
        if label == "as_shipped" and not args.no_spotcheck:
            res[label]["parity_spotcheck"] = spotcheck_first_moves(eng, cfg, net, 0, 0, eng._staggered, ply, args.sims)
        del eng, net
        torch.cuda.empty_cache()
    a, b = res["as_shipped"], res["solver_off"]
    return {"workload": f"the headline batch ({games} concurrent games, 256x10 net, {args.sims} sims/move) with ch5.yml AS SHIPPED: thinking_loop 10, parallel_search_num 8, "
                        f"use_solver_turn 50 (config/ch5.yml:9-16, config.py:133-135,142,154-155); steady-state ply mix, {steps} timed steps after {warm}",
            "value": a["value"], "unit": "sims/s", **{k: v for k, v in a.items() if k not in ("value", "unit")},
            "same_with_the_solver_off": b, "same_with_the_solver_at_the_root_only": res["root_solver_only"],
            "solver_share_of_a_step": {"tree_kernel_ms_with_solver": a["k_tree_par_ms_per_step"], "tree_kernel_ms_without": b["k_tree_par_ms_per_step"],
                                       "tree_kernel_ms_root_solver_only": res["root_solver_only"]["k_tree_par_ms_per_step"],
                                       "share_of_step_time": max(0.0, a["k_tree_par_ms_per_step"] - b["k_tree_par_ms_per_step"]) / a["ms_per_step"]}}


def config1_leg(dev, args, par=1, fused=True, net_kernel=None, shipped=False):
    """BASELINE configs[1]: 4096 concurrent games, mini.yml net, 200 sims/move, WHOLE games (lock-step batch).
    fused: tree and net in ONE kernel, every game's wave evaluating its own leaves (csrc/raz_engine_fused.hip: what the worker runs
    for 16-filter nets); False = the two-kernel pipeline k_tree + k_net_mfma on three streams.
    shipped: mini.yml's play section with no declared override left (config/mini.yml:10-26: thinking_loop 2, parallel_search_num 4,
    end-game solver at the root and inside simulations from turn 50) - only the simulations per move stay BASELINE's 200."""
    import numpy as np
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet, macs_per_position
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    games, sims, chunk = 4096, 200, 200
    cfg = mini_config(sims, par)
    if shipped:
        cfg.play.thinking_loop, cfg.play.use_solver_turn, cfg.play.use_solver_turn_in_simulation = 2, 50, 50
    F, R, V = NETS["mini"]
    blob = ReversiNet(F, R, V).keras_init_(0).to_blob()
    net = DeviceNet(blob, dev, kernel=net_kernel)
    eng = SelfPlayEngine(cfg, net, n_games=games, seed=0, sims_hint=sims * (2 if shipped else 1), fused=fused,
                         solver_budget=int(os.environ.get("RAZ_BENCH_SOLVER_BUDGET", "0")), solver_pool_waves=int(os.environ.get("RAZ_BENCH_SOLVER_WAVES", "0")),
                         solver_pool_every=int(os.environ.get("RAZ_BENCH_POOL_EVERY", "0")), parts=int(os.environ.get("RAZ_BENCH_PARTS", "0")))
    eng.start(0, sims)
    eng.step(50)
    eng.stats()
    eng.start(0, sims)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps, tree_ms, net_ms, timed = 0, 0.0, 0.0, 0
    while True:
        if (steps // chunk) % 8 == 0:
            a, b = eng.step_timed(chunk)
            tree_ms, net_ms, timed = tree_ms + a, net_ms + b, timed + chunk
        else:
            eng.step(chunk)
        steps += chunk
        st = eng.stats()
        if eng.pool_nearly_full(st, chunk):
            eng.gc(min(eng.cfg.nodes_per_game // 4, st["max_pool_used"] // 2))
        if st["finished_games"] >= games:
            break
        if steps > (int(os.environ.get("RAZ_BENCH_MAX_STEPS", "0")) or (80 * sims * 4 + 4000) * (40 if shipped else 1)):   # (a game whose solve is suspended ends its launch: more launches, not more work)
            raise RuntimeError("engine did not finish")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lps = 3
    macs = macs_per_position(F, R, V)
    out = {"fused_tree_net_kernel": bool(fused),
           "workload": f"BASELINE configs[1]: {games} concurrent self-play games/GPU, mini net (F16 R1 V16), {sims} sims/move, mini.yml "
                       + ("play section AS SHIPPED (thinking_loop 2, parallel_search_num 4, end-game solver from turn 50: exact at the root, win/loss inside simulations), "
                          if shipped else f"play settings, thinking_loop=1, solver off, parallel_search_num={par}, ") + "whole games (lock-step batch)"
                       + ("; tree and net in ONE kernel (k_tree_net / k_tree_par_net: the game's wave evaluates its own leaves, up to 256 simulation steps per launch)" if fused else "")
                       + (f"; narrow-net kernel variant {net_kernel}" if net_kernel else ""),
           "value": st["total_sims"] / dt, "unit": "sims/s", "games_per_hour": st["finished_games"] / dt * 3600.0,
           "steps": steps, "ms_per_step": 1e3 * dt / steps, "total_sims": st["total_sims"], "nn_leaves": st["nn_leaves"],
           "finished_games": st["finished_games"], "searched_plies_per_game": st["total_sims"] / games / sims}
    if shipped:
        out["solver_pool"] = dict(eng.solver_stats(), worker_waves=int(eng.cfg.solver_pool_waves) or min(1280, (games + 3) // 4),
                                  iterations_per_round=(int(eng.cfg.reserved) >> 16 & 0xff) * 64 or 96,
                                  what="the end-game solver's pool of worker lanes (csrc/raz_solver_pool.h) over the whole leg: solves, rounds of the pool per answer, "
                                       "share of the worker lanes' iterations spent searching a subtree")
    if fused:
        out["k_tree_net_ms_per_simulation_step"] = tree_ms / timed
        out["matrix_core_tflops_inside_the_fused_kernel"] = 2.0 * macs * st["nn_leaves"] / dt / 1e12
        ach = 2.0 * macs * st["nn_leaves"] / dt / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "k_tree_par_net" if par > 1 else "k_tree_net", "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / FP32_PEAK_TFLOPS, "avg_kernel_ms_per_simulation_step": tree_ms / timed,
                           "note": "net flops of the leaves evaluated / the WHOLE leg's time: the kernel is tree descent + backup + net in one wave per game, so this is "
                                   "the matrix cores' share of a kernel that is bound by one wave's dependent-instruction latency (profiles/r4_pmc/config1_fused_*: "
                                   "matrix pipe busy 38 %, SQ_WAIT_INST_ANY 53 %), not a GEMM efficiency"}
    else:
        leaves_per_launch = st["nn_leaves"] / (steps * lps)
        net_avg = net_ms / (timed * lps)
        ach = 2.0 * macs * leaves_per_launch / (net_avg * 1e-3) / 1e12
        out["leaf_slot_occupancy"] = st["nn_leaves"] / (steps * games * max(par, 1))
        out["roofline"] = {"bound": "mfma", "kernel": "k_net_mfma", "avg_kernel_ms": net_avg, "achieved": ach, "peak": FP32_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS,
                           "note": "3 slices on 3 streams overlap, so the per-launch duration is stretched by co-residency"}
        out["k_tree_avg_ms"] = tree_ms / (timed * lps)
    if (par == 1 or shipped) and not args.no_spotcheck:
        slots = [int(x) for x in np.linspace(0, games - 1, 4 if shipped else 8).astype(int)]
        t1 = time.perf_counter()
        checked = spotcheck_whole_games(eng, cfg, blob, 0, 0, slots, sims)
        out["parity_spotcheck"] = {"result": "ok", "what": "8 game ids sampled from the finished batch: every action and root N == complete oracle games",
                                   "games": checked, "seconds": time.perf_counter() - t1}
    del eng, net
    torch.cuda.empty_cache()
    return out, blob, cfg


def worker_end_to_end_leg(dev, args, seconds=60.0):
    """BatchedSelfPlayWorker.run() itself - what `run.py self` runs - on BASELINE configs[1] (4096 slots, mini net, 200 sims/move) for a
    fixed window: continuous batching inside blocks of 65 536 game ids (worker.start()'s default for a 16-filter net), the block's packed records gathered under an RCCL process group of
    ONE rank (the N > 1 code path), resignation bookkeeping, the native row emitter on host threads and the background writer putting
    play_*.json files on tmpfs (the reference's loop plays, buffers AND writes inside the timed game: worker/self_play.py:139-217,
    lib/data_helper.py:23-25).  games/hour here INCLUDES emission; the engine-level figure of the same blocks is beside it."""
    import glob
    import shutil
    import tempfile
    import torch
    import torch.distributed as dist
    import oracle as O
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker, rows_of_game
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    root = tempfile.mkdtemp(prefix="raz_e2e_", dir=base)
    own_group = not dist.is_initialized()
    # RCCL prints a version banner on the C library's stdout when a communicator is created; this program's stdout carries ONE JSON line,
    # so file descriptor 1 points at stderr while the leg runs (and the C buffers are flushed before it is put back)
    import ctypes
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sims = 200
        mc = mini_config(sims, 1)
        cfg = Config()
        for k, v in vars(mc.play).items():
            setattr(cfg.play, k, v)
        cfg.play.schedule_of_simulation_num_per_move = [[0, sims]]
        cfg.play_data.update(dict(nb_game_in_file=64, nb_game_in_ggf_file=1 << 30, enable_ggf_data=False, max_file_num=24,
                                  save_policy_of_tau_1=bool(mc.play_data.save_policy_of_tau_1), drop_draw_game_rate=0.0))
        rc = cfg.resource
        rc.data_dir, rc.play_data_dir, rc.self_play_ggf_data_dir = root, os.path.join(root, "play"), os.path.join(root, "ggf")
        rc.self_play_game_idx_file = os.path.join(root, ".self-play-game-idx")
        rc.force_simulation_num_file = os.path.join(root, ".force-sim")
        rc.create_directories = lambda: [os.makedirs(d, exist_ok=True) for d in (rc.play_data_dir, rc.self_play_ggf_data_dir)]
        blob = ReversiNet(*NETS["mini"]).keras_init_(0).to_blob()
        from reversi_alpha_zero_amd.worker.self_play import default_block_games
        slots = 4096
        block = default_block_games(blob, slots)   # what worker.start() picks: 16 games per slot for a 16-filter net (65 536 ids, ~9 s per block)
        w = BatchedSelfPlayWorker(cfg, blob, games_in_flight=slots, block_games=block, seed=0, device=str(dev), rank=0, world=1)
        # the first file of the run, for the oracle check below (max_file_num prunes it later)
        keep = {}
        orig_remove = w.remove_play_data

        def remove_and_keep_first(files):
            if files and "first" not in keep:
                keep["first"] = open(files[0], "rb").read()
            return orig_remove(files)
        w.remove_play_data = remove_and_keep_first
        import time as _t
        torch.cuda.synchronize()
        t0 = _t.monotonic()
        w.run(total_games=None, background_emit=True, until=t0 + seconds)
        torch.cuda.synchronize()
        dt = _t.monotonic() - t0
        games = int(open(rc.self_play_game_idx_file).read())
        st = w.last_stats
        blocks = list(getattr(w, "block_stats", []))

        def engine_level(seconds, games_, sims_):   # the engine's own loop: steps + statistics + harvests (host time around them included)
            if not isinstance(seconds, dict):
                return None
            t = max(1e-9, float(seconds.get("steps_and_stats", 0.0)) + float(seconds.get("harvest", 0.0)))
            return {"sims_per_s": sims_ / t, "games_per_hour": games_ / t * 3600.0, "seconds": t}
        out = {"workload": f"BatchedSelfPlayWorker.run() on BASELINE configs[1]: {slots} slots, blocks of {block} game ids with continuous batching, mini net, {sims} sims/move, "
                           "mini.yml play settings (thinking_loop 1, solver off, parallel_search_num 1), RCCL group of one rank for the record gather, native row emitter "
                           f"on {w._emit_executor_threads} host threads + background writer, play_*.json files of 64 games on tmpfs (max_file_num 24)",
               "seconds": dt, "games_written": games, "games_per_hour_including_emission": games / dt * 3600.0,
               "sims_per_s_including_emission": games * st["total_sims"] / max(1, st["finished_games"]) / dt,
               "bytes_written": int(getattr(w, "bytes_written", 0)), "gb_per_s_of_json_text": getattr(w, "bytes_written", 0) / dt / 1e9,
               "writer_busy_share_of_the_run": w.last_writer_busy_seconds / dt, "blocks": len(blocks), "pieces_handed_to_the_writer": w.last_writer_batches,
               "streamed_emission": "the finished prefix of a block goes to the writer while the block is played (worker.run: one rank, exact-f32 net)",
               "main_thread_seconds": dict(getattr(w, "run_seconds", {}) or {}),
               "engine_level_of_the_last_block": engine_level(st.get("seconds"), st["finished_games"], st["total_sims"]),
               "engine_level_of_all_blocks": engine_level({k: sum(b.get(k, 0.0) for b in blocks) for k in ("steps_and_stats", "harvest")},
                                                          sum(b["games"] for b in blocks), sum(b["sims"] for b in blocks)),
               "blocks_detail": blocks[:3] + blocks[-2:],
               "gather_backend": getattr(w, "last_gather_backend", None), "gather_bytes_last_block": getattr(w, "last_gather_bytes", None),
               "host_threads": os.cpu_count()}
        # two games of the first file against the oracle's rows for the same ids
        first = json.loads(keep.get("first") or open(sorted(glob.glob(os.path.join(rc.play_data_dir, "*.json")))[0], "rb").read())
        ocfg = O.play_cfg_from_config(mc)
        at, checked = 0, []
        for gid in (0, 1):
            plies, summ = O.selfplay_game(ocfg, blob, 0, gid, sims)
            rows = json.loads(json.dumps(rows_of_game(plies, summ["winner"])))
            ok = first[at:at + len(rows)] == rows
            checked.append({"game_id": gid, "rows": len(rows), "ok": bool(ok)})
            at += len(rows)
            if not ok:
                raise AssertionError(f"worker_end_to_end: the rows of game {gid} in the first play_*.json differ from the oracle's")
        out["parity_check_files"] = {"result": "ok", "what": "the rows of game ids 0 and 1 in the run's first play_*.json == the rows of the games the CPU oracle plays for those ids", "games": checked}
        w._drop_engine(net_too=True)
        return out
    finally:
        if own_group:
            dist.destroy_process_group()
        shutil.rmtree(root, ignore_errors=True)
        try:
            ctypes.CDLL(None).fflush(None)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)


def continuous_leg(dev, args, rounds=3, shipped=False):
    """shipped: mini.yml's play section as shipped (thinking_loop 2, parallel_search_num 4, solver from turn 50) - the steady state of
    the solver-bound configuration: slots are at every stage of a game at once, so the solver pool's lanes are shared between the
    few games that are solving instead of waiting for the slowest game of a lock-step batch.
    configs[1] with continuous batching (raz_engine_harvest: a finished slot restarts on the next game id at once,
    worker/self_play.py:95-137): 4096 slots, rounds x 4096 game ids, whole games.  games/hour and sims/s here are measured
    on complete games in a (mostly) steady state, not extrapolated; leaf_slot_occupancy = nn leaves / (steps x slots)."""
    import numpy as np
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine, raw_from_packed
    games, sims = 4096, 200
    cfg = mini_config(sims, 4 if shipped else 1)
    if shipped:
        cfg.play.thinking_loop, cfg.play.use_solver_turn, cfg.play.use_solver_turn_in_simulation = 2, 50, 50
    blob = ReversiNet(*NETS["mini"]).keras_init_(0).to_blob()
    eng = SelfPlayEngine(cfg, DeviceNet(blob, dev), n_games=games, seed=0, sims_hint=sims * (2 if shipped else 1), fused=False,
                         solver_budget=int(os.environ.get("RAZ_BENCH_SOLVER_BUDGET", "0")), solver_pool_waves=int(os.environ.get("RAZ_BENCH_SOLVER_WAVES", "0")),
                         solver_pool_every=int(os.environ.get("RAZ_BENCH_POOL_EVERY", "0")), parts=int(os.environ.get("RAZ_BENCH_PARTS", "0")))
    eng.start(0, sims)
    eng.step(50)
    eng.stats()
    torch.cuda.synchronize()
    outbox, st = eng.play_continuous(0, rounds * games, lambda gid: sims, chunk=200)
    torch.cuda.synchronize()
    # the stepping loop (steps + per-chunk stats + harvests, host clock, every chunk ends synchronised) - like the lock-step
    # leg, the start of the batch and the allocation of the id-ordered outbox are outside the timed region
    dt = st["seconds"]["steps_and_stats"] + st["seconds"]["harvest"]
    out = {"workload": f"BASELINE configs[1] with continuous batching: {games} slots, {rounds * games} game ids (whole games), mini net, {sims} sims/move, "
                       + ("mini.yml play section AS SHIPPED (thinking_loop 2, parallel_search_num 4, end-game solver from turn 50), two-kernel pipeline + solver pool"
                          if shipped else "mini.yml play settings, thinking_loop=1, solver off, parallel_search_num=1"),
           "value": st["total_sims"] / dt, "unit": "sims/s", "games_per_hour": st["finished_games"] / dt * 3600.0,
           "finished_games": st["finished_games"], "steps": st["steps"], "ms_per_step": 1e3 * dt / st["steps"],
           "leaf_slot_occupancy": st["leaf_slot_occupancy"], "gc_runs": st["gc_runs"], "seconds": st["seconds"],
           "note": "includes the ramp-down of the last games (slots idle once no unplayed id is left) and one host synchronisation per 200 steps"}
    if not args.no_spotcheck:
        import oracle as O
        import concurrent.futures as cf
        ids = [int(x) for x in np.linspace(0, rounds * games - 1, 8).astype(int)]
        rows = torch.tensor(ids, device=dev)
        raw = raw_from_packed(*(outbox[k][rows].cpu().numpy() for k in ("headers", "root_n", "summary")))
        ocfg = O.play_cfg_from_config(cfg, parallel_search_num=4 if shipped else 1)
        if shipped:
            out["solver_pool"] = eng.solver_stats()
        with cf.ThreadPoolExecutor(max_workers=8) as ex:
            ref = list(ex.map(lambda gid: O.selfplay_game(ocfg, blob, 0, gid, sims), ids))
        for r, (gid, (plies, summ)) in enumerate(zip(ids, ref)):
            n = int(raw["n_plies"][r])
            ok = int(raw["game_id"][r]) == gid and [int(a) for a in raw["headers"][r, :n]["action"]] == [p["action"] for p in plies] \
                and all([float(x) for x in raw["root_n"][r, i]] == p["root_n"] for i, p in enumerate(plies))
            if not ok:
                raise AssertionError(f"parity spot check FAILED: continuous batching, game id {gid} differs from the oracle")
        out["parity_spotcheck"] = {"result": "ok", "what": "8 game ids sampled from the id-ordered outbox (refilled slots included): every action and root N == complete oracle games",
                                   "game_ids": ids}
    del eng
    torch.cuda.empty_cache()
    return out


def sweep_leg(dev):
    """The bitboard-sweep HBM leg of the north star: k_step / k_legal_moves GB/s (tools/bench_sweep.py) at 2^24 boards (the
    size SURVEY 8(d) names; its 256 MiB of inputs equal the Infinity Cache) and at 2^26 boards (1 GiB of inputs: nothing is
    cache-resident), with the counter traffic of the committed PMC passes beside the algorithmic bytes."""
    import bench_sweep
    tr = sweep_traffic()
    res = {}
    for boards, steps in ((1 << 24, 10), (1 << 26, 4)):
        a = types.SimpleNamespace(boards=boards, steps=steps, warmup=2, no_cpu_baseline=True, cpu_budget=0.0)
        o = bench_sweep.run_sweep(a, 0, 1, dev)
        ks, kl = o["roofline"], o["k_legal_moves"]
        for k, name in ((ks, "k_step"), (kl, "k_legal_moves")):
            t = tr.get(f"{name}@{boards}")
            if t:
                # FETCH_SIZE counts the 128-byte requests of wide (16 B/lane) coalesced reads at 64 B on gfx950 and is "uncalibrated" for
                # other widths (MI355X_MICROARCH.md): these kernels read 8 B/lane, and their inputs are a known byte count that must
                # be fetched at least once - so the raw figure is used where it covers the algorithmic reads, the doubled one where not
                read_alg = {"k_step": 19, "k_legal_moves": 16}[name] * boards
                doubled = t["fetch_bytes_raw"] < 0.98 * read_alg
                k["traffic"] = (2.0 if doubled else 1.0) * t["fetch_bytes_raw"] + t["write_bytes"]
                k["traffic_over_algorithmic"] = k["traffic"] / k["algorithmic_bytes_per_launch"]
                k["traffic_detail"] = {"fetch_bytes_raw": t["fetch_bytes_raw"], "fetch_doubled": doubled, "write_bytes": t["write_bytes"],
                                       "algorithmic_read_bytes": read_alg, "dispatches_averaged": t.get("dispatches_averaged")}
                k["traffic_source"] = f"profiles/{PMC_DIR}/sweep_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/bench_sweep.py on this round's kernels)"
        res[f"boards_2^{boards.bit_length() - 1}"] = {"workload": o["config"]["workload"], "k_step": ks, "k_legal_moves": kl, "boards_per_s": o["value"]}
    out = dict(res["boards_2^24"])
    out["beyond_the_infinity_cache_2^26_boards"] = res["boards_2^26"]
    return out


def cpu_baseline_port(cfg, blob, sims, threads, stop_after_plies=0, what=""):
    """The oracle (C port of agent/player.py + env + bitboard + the same net), one game per host thread (ctypes
    releases the GIL), same settings.  Reported, not optimised."""
    import concurrent.futures as cf
    import oracle as O
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=cfg.play.parallel_search_num)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        res = list(ex.map(lambda gid: O.selfplay_game(ocfg, blob, 0, gid, sims, stop_after_plies=stop_after_plies)[1], range(threads)))
    dt = time.perf_counter() - t0
    total = sum(r["n_sims"] for r in res)
    return {"value": total / dt, "unit": "sims/s", "cores": threads, "kind": "port",
            "sample": f"{what}; oracle/orc_mcts.c + orc_net.c (C port of the reference player and net), one game per host thread; "
                      f"{total} sims in {dt:.1f} s"}


# ------------------------------------------------------------------------------------------------------------
def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) started without a launcher: become N ranks."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("RAZ_BENCH_SHARED_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible (RAZ_BENCH_SHARED_GPU=1 runs the ranks on shared "
                         f"GPUs over gloo: a test rig, labelled as such)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def compact_line(full):
    """The ONE line the driver records (<= 8 KB): the contract keys, `roofline`, `cpu_baseline`, and one short object per leg with its
    value and its parity result.  Everything else (per-leg detail, notes, game lists) is in the full document (bench_full.json)."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else d

    def parity(d):
        if not isinstance(d, dict):
            return d
        return {"result": d.get("result"), "games": len(d.get("games", d.get("game_ids", [])))}
    line = pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = pick(full.get("config", {}), ("workload", "games_per_gpu", "sims_per_move", "net", "test_rig", "collective_backend"))
    r = full.get("roofline", {})
    line["roofline"] = pick(r, ("bound", "kernel_name", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "algorithmic_flops_per_launch",
                                "algorithmic_hbm_bytes_per_launch", "executed_mfma_tflops", "executed_mfma_frac_of_pipe_peak", "frac_of_f32_mfma_peak",
                                "power_limited"))
    c = full.get("cpu_baseline")
    if isinstance(c, dict):
        line["cpu_baseline"] = pick(c, ("value", "unit", "cores", "kind", "sample"))
        if isinstance(c.get("configs1_mini_200sims"), dict):
            line["cpu_baseline"]["configs1_value"] = c["configs1_mini_200sims"].get("value")
    line["games_per_hour"] = full.get("games_per_hour")
    line["leaves_per_sec"] = full.get("leaves_per_sec")
    line["parity_spotcheck"] = parity(full.get("parity_spotcheck"))
    line["parity_spotcheck_timed_batch"] = parity(full.get("parity_spotcheck_timed_batch"))
    nc = full.get("net_check")
    line["net_check"] = (dict(pick(nc, ("n", "max_abs_dp", "max_abs_dv", "tolerance", "result")), vs_f64_sample=nc.get("sample_vs_the_graph_in_f64"))
                         if isinstance(nc, dict) else nc)
    w = full.get("whole_games_measured")
    if isinstance(w, dict):
        line["whole_games_measured"] = dict(pick(w, ("value", "unit", "seconds", "games_per_hour", "games_finished_in_the_window", "leaf_slot_occupancy",
                                                     "sims_per_net_evaluation", "ms_per_step")),
                                            slots=full.get("config", {}).get("games_per_gpu"), parity=parity(w.get("parity_check_complete_games")))
        line["whole_games_measured"]["sims_per_s"] = w.get("value")
        if isinstance(w.get("emission"), dict):   # the worker's files written inside the window
            line["whole_games_measured"]["emission"] = pick(w["emission"], ("games_written", "bytes_written", "writer_busy_share_of_the_window",
                                                                             "games_per_hour_including_emission", "drain_after_the_window_seconds"))
    for key in ("headline_on_exact_f32_kernels", "ch5_yml_as_shipped", "config5_8192x3200_agz", "config1_4096x200_mini", "config1_mini_yml_parallel_search_num_4", "config1_mini_yml_as_shipped", "config1_mini_yml_as_shipped_two_kernel_pipeline",
                "config1_two_kernel_pipeline", "config1_two_kernel_pipeline_parallel_search_num_4", "config1_continuous_batching",
                "config1_mini_yml_as_shipped_continuous_batching"):
        d = full.get(key)
        if isinstance(d, dict):
            e = pick(d, ("value", "unit", "games_per_hour", "ms_per_step", "error", "fused_tree_net_kernel"))
            if "parity_spotcheck" in d:
                e["parity"] = parity(d["parity_spotcheck"])
            if "roofline" in d:
                e["roofline_frac"] = d["roofline"].get("frac")
            if "solver_share_of_a_step" in d:
                e["solver_share_of_step_time"] = d["solver_share_of_a_step"].get("share_of_step_time")
                e["solver_off_value"] = d.get("same_with_the_solver_off", {}).get("value")
                e["root_solver_only_value"] = d.get("same_with_the_solver_at_the_root_only", {}).get("value")
                po = (d.get("parity_spotcheck") or {}).get("played_on") if isinstance(d.get("parity_spotcheck"), dict) else None
                if po:
                    e["played_on_steps"], e["played_on_sims_per_s"] = po.get("steps"), po.get("sims_per_s")
            if "parity_check_complete_games" in d:
                e["parity"] = parity(d["parity_check_complete_games"])
            if isinstance(d.get("solver_pool"), dict):
                e["solver_pool"] = pick(d["solver_pool"], ("solves", "pool_rounds_per_answer", "lane_utilisation", "most_rounds_listed_of_one_game", "worker_waves"))
            line[key] = e
    d = full.get("worker_end_to_end_config1")
    if isinstance(d, dict):
        line["worker_end_to_end_config1"] = pick(d, ("games_written", "seconds", "games_per_hour_including_emission", "sims_per_s_including_emission", "bytes_written",
                                                     "writer_busy_share_of_the_run", "gather_backend", "main_thread_seconds", "engine_level_of_the_last_block", "engine_level_of_all_blocks", "error"))
        if "parity_check_files" in d:
            line["worker_end_to_end_config1"]["parity"] = parity(d["parity_check_files"])
    sw = full.get("bitboard_sweep")
    if isinstance(sw, dict) and "k_step" in sw:
        line["bitboard_sweep"] = {k: pick(sw[k], ("achieved", "peak", "unit", "frac", "traffic_over_algorithmic")) for k in ("k_step", "k_legal_moves")}
        b26 = sw.get("beyond_the_infinity_cache_2^26_boards", {})
        line["bitboard_sweep"]["frac_at_2^26_boards"] = {k: b26.get(k, {}).get("frac") for k in ("k_step", "k_legal_moves")}
    elif isinstance(sw, dict):
        line["bitboard_sweep"] = sw
    for key in ("record_gather", "config4_acceptance"):
        if key in full:
            line[key] = pick(full[key], ("collective", "bytes", "games", "seconds", "payload_check", "result", "gathered_bytes"))
    line["bench_wall_seconds"] = full.get("bench_wall_seconds")
    line["full_document"] = full.get("_full_path")

    def sig(x):   # seven significant digits say everything a measurement here can (the full document keeps the doubles): ~10 % of the line
        if isinstance(x, float) and x == x and abs(x) not in (0.0, float("inf")):
            return float(f"{x:.7g}")
        if isinstance(x, dict):
            return {k: sig(v) for k, v in x.items()}
        if isinstance(x, list):
            return [sig(v) for v in x]
        return x
    return sig(line)


T_MAIN = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--games", type=int, default=8192, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--net", default="ch5", choices=sorted(NETS))
    ap.add_argument("--net-kernel", default="auto", help="DeviceNet kernel: auto = f16x3 (raznet-forward-v2) where supported, f32 = exact-f32 kernels")
    ap.add_argument("--parts", type=int, default=0, help="slices/streams the batch is stepped in (0 = 1 for the 256x10 net, 3 for mini)")
    ap.add_argument("--nodes-per-game", type=int, default=0, help="node pool per game (0 = 16 x sims: pools are pruned by k_gc in long runs)")
    ap.add_argument("--tree-warm", type=int, default=40, help="untimed steps on the staggered batch before the W warm-up steps")
    ap.add_argument("--opening", action="store_true", help="time the first steps from the opening instead of the steady state")
    ap.add_argument("--no-spotcheck", action="store_true")
    ap.add_argument("--no-leaf-cache", action="store_true", help="headline engine without the cross-game evaluation cache")
    ap.add_argument("--fused", action="store_true", help="--net mini only: the headline leg on the fused tree + net kernel (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline only (every other leg skipped)")
    ap.add_argument("--no-whole-games", action="store_true", help="skip the fixed-window leg on the headline configuration (~2.5 min)")
    ap.add_argument("--window-seconds", type=float, default=120.0, help="length of the fixed window the headline configuration is played for")
    ap.add_argument("--legs", default=None, help="comma-separated keys: run only these extra legs (e.g. ch5_yml_as_shipped,config1_4096x200_mini)")
    ap.add_argument("--time-budget", type=float, default=600.0,
                    help="seconds: an extra leg is not STARTED when the run's elapsed time plus the leg's nominal duration would pass this "
                         "(the default run takes ~7 min on an MI355X box; on a slower box legs are dropped from the end, each with a note, "
                         "instead of the line arriving late or not at all)")
    ap.add_argument("--full-out", default=None, help="where the full document goes (default: gpurun_out/bench_full.json if gpurun_out/ exists, else ./bench_full.json)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(args)
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path is device-only; there is no CPU fallback)")
    # RAZ_BENCH_SHARED_GPU=1 (test rig only): several ranks share the visible GPUs and rendezvous over
    # gloo, to exercise the N > 1 code path on a 1-GPU box; the reported line says so.
    # RAZ_BENCH_NCCL_WORLD1=1 (test rig only): at N = 1, join an nccl (RCCL) group of ONE rank and run everything the N > 1 path runs
    # on it - barriers, the all-reduces of the timed region, the record gather from HBM with its payload check.
    shared_gpu = os.environ.get("RAZ_BENCH_SHARED_GPU") == "1"
    nccl_world1 = world == 1 and os.environ.get("RAZ_BENCH_NCCL_WORLD1") == "1"
    if shared_gpu:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if shared_gpu else dev   # where the collectives' tensors live
    if world > 1 or nccl_world1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:   # (a launcher always sets it; ranks started by hand cannot agree on a port nobody named)
                raise SystemExit("bench.py: WORLD_SIZE > 1 needs MASTER_PORT (torch.distributed.run sets it; `python bench.py --gpus N` picks a free one)")
            import socket
            with socket.socket() as sock:   # a group of one rank: any free port (29511 may belong to another run on a shared box)
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    group = world > 1 or nccl_world1   # the collective code path runs
    import __graft_entry__ as g
    if world > 1:   # one rank builds (a no-op when the shipped .so files are current), the others wait
        if local == 0:
            g.build()
        dist.barrier()
    g.build()

    acceptance = None
    if group and (not args.no_extra_legs or nccl_world1):
        acceptance = config4_acceptance(dev, rank, world, cdev)
    res = headline_leg(args, dev, rank, world, cdev, group)
    if rank == 0:
        out, blob, cfg = res
        ply_weights = out.pop("_ply_weights", None)
        if acceptance:
            out["config4_acceptance"] = acceptance
        if shared_gpu:
            out["config"]["test_rig"] = "RAZ_BENCH_SHARED_GPU=1: ranks share GPUs, gloo collectives - not a scaling measurement"
        if nccl_world1:
            out["config"]["test_rig"] = "RAZ_BENCH_NCCL_WORLD1=1: the N > 1 code path (barriers, all-reduces, record gather) on an RCCL group of one rank"
        import gc
        checks = None
        if world == 1 and not args.no_extra_legs and args.net == "ch5" and not args.no_whole_games:
            gc.collect()
            torch.cuda.empty_cache()
            wg, recs = steady_window_leg(args, dev, blob, cfg, ply_weights, slots=args.games, seconds=args.window_seconds)
            out["whole_games_measured"] = wg
            # the line's games/hour is what was measured over the window; the 20-step extrapolation stays beside it
            out["games_per_hour_extrapolated_from_the_timed_steps"] = out["games_per_hour"]
            out["games_per_hour"] = wg["games_per_hour"]
            out["games_per_hour_note"] = (f"MEASURED in this run: games finished inside a {wg['seconds']:.0f} s window of the headline configuration played with continuous "
                                          "batching (whole_games_measured)")
            out["headline_over_whole_games_sims_per_s"] = out["value"] / wg["value"]
            if not args.no_spotcheck and recs:
                checks = start_complete_game_checks(dev, args, blob, cfg, recs)
        if world == 1 and not args.no_cpu_baseline:
            # the reference's own pure-Python self-play on this box's host cores, in this run (the game checks above keep two host
            # threads and an otherwise idle GPU busy meanwhile)
            ref = cpu_baseline_reference()
            if ref is not None:
                ref["sample"] = (ref.get("sample", "") + f"; the reference runs its yml's parallel_search_num 8 (throughput setting, SURVEY 8(d)), the GPU headline "
                                 f"parallel_search_num 1 - ch5_yml_as_shipped is the GPU figure at 8").strip("; ")
                out["cpu_baseline"] = ref
            else:   # no reference modules on this box (oracle/_ref not staged): the C port, labelled as such
                threads = min(os.cpu_count() or 1, 128)
                per_thread = 2 if args.net == "ch5" else 400
                out["cpu_baseline"] = cpu_baseline_port(
                    cfg, blob, per_thread, threads, stop_after_plies=2,
                    what=f"oracle/_ref absent: C port instead of the reference; bounded sample: the first {per_thread} simulations of the first searched "
                         f"move of {threads} games")
        if checks is not None:
            out["whole_games_measured"]["parity_check_complete_games"] = join_complete_game_checks(
                *checks, what="games of the window that were taken up late (ply >= 50) and finished inside it: every action, the winner and every ply's root N == "
                              "the game the CPU oracle plays for that id from the same position with the device net's outputs (800 sims/move)")
        if world == 1 and not args.no_extra_legs:
            legs = ((("headline_on_exact_f32_kernels", lambda: exact_f32_leg(args, dev, blob, cfg, ply_weights)),
                     ("ch5_yml_as_shipped", lambda: ch5_as_shipped_leg(args, dev, blob, ply_weights, games=args.games)))
                    if "f16x3" in out["dtype"] else ()) + (
                    ("config5_8192x3200_agz", lambda: config5_leg(args, dev, blob, ply_weights)),
                    ("config1_4096x200_mini", lambda: config1_leg(dev, args, 1, fused=True)[0]),
                    ("config1_mini_yml_parallel_search_num_4", lambda: config1_leg(dev, args, 4, fused=True)[0]),
                    ("config1_mini_yml_as_shipped", lambda: config1_leg(dev, args, 4, fused=True, shipped=True)[0]),
                    ("config1_mini_yml_as_shipped_two_kernel_pipeline", lambda: config1_leg(dev, args, 4, fused=False, shipped=True)[0]),
                    ("config1_two_kernel_pipeline", lambda: config1_leg(dev, args, 1, fused=False)[0]),
                    ("config1_two_kernel_pipeline_parallel_search_num_4", lambda: config1_leg(dev, args, 4, fused=False)[0]),
                    ("config1_continuous_batching", lambda: continuous_leg(dev, args)),
                    ("config1_mini_yml_as_shipped_continuous_batching", lambda: continuous_leg(dev, args, rounds=3, shipped=True)),
                    ("worker_end_to_end_config1", lambda: worker_end_to_end_leg(dev, args)),
                    ("bitboard_sweep", lambda: sweep_leg(dev)))
            only = set(args.legs.split(",")) if args.legs else None
            # nominal seconds of a leg on an MI355X box (profiles/r5/bench_r5_default_run_*: engine construction included)
            nominal = {"worker_end_to_end_config1": 85.0, "config5_8192x3200_agz": 35.0, "ch5_yml_as_shipped": 30.0, "headline_on_exact_f32_kernels": 20.0,
                       "config1_mini_yml_as_shipped_continuous_batching": 20.0}
            for key, leg in legs:
                if only is not None and key not in only:
                    continue
                elapsed = time.perf_counter() - T_MAIN
                if only is None and elapsed + nominal.get(key, 10.0) > args.time_budget:
                    out[key] = {"error": f"not run: {elapsed:.0f} s into the run, --time-budget {args.time_budget:.0f} s"}
                    continue
                gc.collect()
                torch.cuda.empty_cache()
                try:
                    out[key] = leg()
                except AssertionError:
                    raise
                except Exception as ex:   # never lose the main line over an extra leg
                    out[key] = {"error": repr(ex)}
        out["bench_wall_seconds"] = time.perf_counter() - T_MAIN
        path = args.full_out or os.path.join(ROOT, "gpurun_out" if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "", "bench_full.json")
        try:
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            out["_full_path"] = os.path.relpath(path, ROOT)
        except OSError:
            out["_full_path"] = None
        print(json.dumps(compact_line(out)))
    if group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
