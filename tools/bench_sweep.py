#!/usr/bin/env python
"""tools/bench_sweep.py — HBM roofline of the bitboard sweep kernels (the "bitboard sweep" leg of the
north star); the headline bench is /bench.py.  One JSON line on stdout.

    python tools/bench_sweep.py [--steps K] [--warmup W] [--boards N]

Workload `sweep` (bitboard sweep leg of BASELINE.json's north star): one step = one
`raz_step_batch` pass (ReversiEnv.step: flip + place + pass/terminal + next legal mask) over
2^24 positions harvested from random self-play at uniformly random ply (SURVEY §8(d) "value
distributions"), inputs resident in HBM, each step on its own copy of the batch (755 MB, larger
than the 256 MiB Infinity Cache, so no step re-reads cached lines).

Multi-GPU: positions shard across ranks with no data-path collective (weak scaling); the timed
region is bracketed by barrier + synchronize and the MAX over ranks is reported.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (≈6.3 TB/s measured float4 copy)
STEP_BYTES_PER_BOARD = 45  # include/raz.h raz_step_batch: 19 B read + 26 B written


def harvest_positions(n, seed, dev, ply_weights=None):
    """Random playouts on device; game g is frozen at a ply drawn from [0, 58]: uniformly, or with the given 59 weights."""
    import torch
    from reversi_alpha_zero_amd.lib import bitboard as bb
    black = torch.full((n,), 0x0000000810000000, dtype=torch.int64, device=dev)
    white = torch.full((n,), 0x0000001008000000, dtype=torch.int64, device=dev)
    player = torch.ones(n, dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.uint8, device=dev)
    legal = bb.legal_moves_batch(black, white)
    g = torch.Generator(device=dev).manual_seed(seed)
    if ply_weights is None:
        target = torch.randint(0, 59, (n,), generator=g, device=dev, dtype=torch.int32)
    else:
        w = torch.as_tensor(ply_weights, dtype=torch.float32, device=dev)
        target = torch.multinomial(w / w.sum(), n, replacement=True, generator=g).to(torch.int32)
    snap = [black.clone(), white.clone(), player.clone(), legal.clone()]
    for ply in range(59):
        take = (target == ply) & (status == 0)
        for s, cur in zip(snap, (black, white, player, legal)):
            s.copy_(torch.where(take, cur, s))
        rnd = torch.randint(0, 2**31 - 1, (n,), generator=g, device=dev, dtype=torch.int32)
        action = bb.pick_kth_legal_batch(legal, rnd)
        bb.step_batch(black, white, player, status, legal, action)
    rnd = torch.randint(0, 2**31 - 1, (n,), generator=g, device=dev, dtype=torch.int32)
    action = bb.pick_kth_legal_batch(snap[3], rnd)
    return snap[0], snap[1], snap[2], action


def cpu_baseline_sweep(black, white, player, action, budget_s=10.0):
    """Oracle (C port of env/reversi_env.py step) on one host core over a bounded sample."""
    import numpy as np
    import oracle
    lib = oracle.load()
    m = min(black.numel(), 1 << 21)
    b = black[:m].cpu().numpy().view(np.uint64).copy()
    w = white[:m].cpu().numpy().view(np.uint64).copy()
    p = player[:m].cpu().numpy().copy()
    a = action[:m].cpu().numpy().copy()
    done_boards, acc, t0 = 0, 0.0, time.perf_counter()
    while True:
        bb_, ww, pp = b.copy(), w.copy(), p.copy()
        st = np.zeros(m, dtype=np.uint8)
        lg = np.zeros(m, dtype=np.uint64)
        t1 = time.perf_counter()
        lib.orc_step_n(bb_.ctypes.data, ww.ctypes.data, pp.ctypes.data, st.ctypes.data, lg.ctypes.data,
                       a.ctypes.data, m)
        dt = time.perf_counter() - t1
        done_boards += m
        acc += dt
        if time.perf_counter() - t0 > budget_s:
            break
    return {"value": done_boards / acc, "unit": "boards/s", "cores": 1, "kind": "port",
            "sample": f"{done_boards} env.step calls on harvested self-play positions (oracle/orc_bitboard.c, 1 thread)"}


def run_sweep(args, rank, world, dev):
    import torch
    import torch.distributed as dist
    from reversi_alpha_zero_amd.lib import bitboard as bb
    n = args.boards
    black, white, player, action = harvest_positions(n, 12345 + rank, dev)
    copies = args.steps + args.warmup
    sets = []
    for _ in range(copies):
        sets.append((black.clone(), white.clone(), player.clone(), torch.zeros(n, dtype=torch.uint8, device=dev),
                     torch.empty(n, dtype=torch.int64, device=dev)))
    torch.cuda.synchronize()
    for i in range(args.warmup):
        b, w, p, s, l = sets[i]
        bb.step_batch(b, w, p, s, l, action)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        b, w, p, s, l = sets[args.warmup + i]
        evs[i][0].record()
        bb.step_batch(b, w, p, s, l, action)
        evs[i][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps
    total_boards = n * world * args.steps
    out = {
        "metric": "bitboard sweep: ReversiEnv.step positions/sec (sub-metric of MCTS sims/sec/GPU)",
        "value": total_boards / elapsed, "unit": "boards/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"raz_step_batch over {n} positions/GPU harvested from random self-play "
                               f"at uniform random ply (SURVEY §8(d))", "boards_per_gpu": n},
        "roofline": {"bound": "hbm", "achieved": STEP_BYTES_PER_BOARD * n / (kern_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                     "kernel": step_kernel_name(n), "avg_kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": STEP_BYTES_PER_BOARD * n},
    }
    out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
    # the mobility kernel alone (24 B/board)
    lg = torch.empty(n, dtype=torch.int64, device=dev)
    for _ in range(3):
        bb.legal_moves_batch(black, white, out=lg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        bb.legal_moves_batch(black, white, out=lg)
    e1.record()
    torch.cuda.synchronize()
    lm_ms = e0.elapsed_time(e1) / 10
    sliced = bb.sweep_forms(n)[0] == 1
    out["k_legal_moves"] = {"bound": "hbm", "kernel": ("k_legal_moves_sliced (32 boards per lane, bit-sliced: csrc/raz_sweep_sliced.h; batches from 2^25 boards on)"
                                                       if sliced else "k_legal_moves (a board per lane)"), "avg_kernel_ms": lm_ms, "achieved": 24 * n / (lm_ms * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "algorithmic_bytes_per_launch": 24 * n,
                            "frac": 24 * n / (lm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": f"the same {16 * n >> 20} MiB of inputs every launch: " + ("Infinity-Cache (256 MiB) resident in part" if 16 * n <= (512 << 20)
                                                                                              else "4x the 256 MiB Infinity Cache, streamed from HBM")}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_sweep(black, white, player, action, args.cpu_budget)
    return out


def step_kernel_name(n):
    """Which kernel raz_step_batch runs the whole superblocks of a batch of n boards on (the library's own answer: raz_sweep_forms)."""
    from reversi_alpha_zero_amd.lib import bitboard as bb
    return {0: "k_step (a board per lane)", 1: "k_step_sliced (32 boards per lane, everything bit-sliced)",
            2: "k_step_hybrid (32 boards per lane: the move per board, the legal moves after it bit-sliced; batches from 2^26 boards on)"}[bb.sweep_forms(n)[1]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sweep", choices=["sweep"])
    ap.add_argument("--boards", type=int, default=1 << 24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path is device-only; there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import __graft_entry__ as g
    g.build()
    out = run_sweep(args, rank, world, dev)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
