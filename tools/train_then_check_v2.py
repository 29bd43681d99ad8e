#!/usr/bin/env python
"""tools/train_then_check_v2.py - raznet-forward-v2 (the split-f16 trunk, csrc/raz_net_f16x3.hip) on a net that has been TRAINED,
on a machine without a GPU.

Every net that has run here is a random initialisation (no checkpoint travels with the repository): the 1e-5 bound
of v2 against fp32 is accepted on that evidence.  This tool makes the evidence one step less synthetic with what a CPU can do:
  1. self-play games by the CPU oracle (oracle/orc_mcts.c, mini net, a few simulations per move) give training rows the way the
     reference's worker does - [own, enemy] planes, the root's visit distribution, z (worker/self_play.py:180-194);
  2. the 256x10 net (agent/model.py:28-72) is trained on them with the reference's recipe (worker/optimize.py:72-111: SGD momentum
     0.9, categorical cross-entropy + mean squared error, l2 1e-4; BatchNorm momentum 0.99 as in Keras) for --steps steps of --batch
     rows on the host cores - BatchNorm statistics and weights are then those of a net in training, not of an initialisation;
  3. the net's blob is evaluated on held-out game positions by (a) the fp32 torch graph (the reference for the tolerance), (b) the
     same graph in f64, (c) the PRODUCT's kernels compiled for the host by the wave emulator (tests/native/libraz_emu_net.so: the
     same source as libraz.so, f16 matrix instructions emulated by their lane layouts) - exact-f32 kernels (v1) and split-f16
     kernels (v2).
One JSON document.  The emulator runs the kernels' arithmetic in the order the lane layouts prescribe; what it cannot show is what
the hardware adds (nothing, for these instructions: tools/probe_f16.hip) - the GPU figures of record are
profiles/r3/net_accuracy_*.json (131 072 positions, five random-init variants) and tests/test_engine_gpu.py.
    python tools/train_then_check_v2.py --steps 2500 --out profiles/r5/net_v2_on_a_cpu_trained_256x10_net_wave_emulator.json"""
import argparse
import ctypes
import json
import os
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def rows_from_oracle_games(n_games, sims, seed, threads):
    """[(own, enemy, policy64, z)] of every searched ply of n_games oracle games (mini net of seed 0, mini.yml-like play settings)."""
    import concurrent.futures as cf
    import oracle as O
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.config import Config
    cfg = Config()
    cfg.play.thinking_loop, cfg.play.parallel_search_num = 1, 1
    cfg.play.use_solver_turn = cfg.play.use_solver_turn_in_simulation = 0
    cfg.play.resign_threshold = None
    ocfg = O.play_cfg_from_config(cfg)
    blob = ReversiNet(16, 1, 16).keras_init_(0).to_blob()

    def one(g):
        plies, summ = O.selfplay_game(ocfg, blob, seed, g, sims)
        w = summ["winner"]
        black_win = 1 if w == 1 else (-1 if w == 2 else 0)
        out = []
        for p in plies:
            if not p["has_row"]:
                continue
            n = np.asarray(p["root_n"], dtype=np.float64)
            if n.sum() <= 0:
                continue
            out.append((p["own"], p["enemy"], (n / n.sum()).astype(np.float32), black_win if p["player"] == 1 else -black_win))
        return out
    with cf.ThreadPoolExecutor(threads) as ex:
        games = list(ex.map(one, range(n_games)))
    return [r for g in games for r in g]


def planes_of(own, enemy):
    sh = np.arange(64, dtype=np.uint64)
    o = ((np.asarray(own, dtype=np.uint64)[:, None] >> sh) & np.uint64(1)).astype(np.float32)
    e = ((np.asarray(enemy, dtype=np.uint64)[:, None] >> sh) & np.uint64(1)).astype(np.float32)
    return np.stack([o, e], axis=1).reshape(-1, 2, 8, 8)


def emu_lib():
    import fcntl
    emu_dir = os.path.join(ROOT, "tests", "native", "wave_emu")
    with open(os.path.join(ROOT, "tests", "native", ".emu.lock"), "w") as lock:   # (the tests' lock: one build of the emulator libraries at a time)
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-C", emu_dir, "../libraz_emu_net.so"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-2000:] + r.stderr[-2000:])
    from reversi_alpha_zero_amd import _native as N
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "native", "libraz_emu_net.so"))
    lib.raz_last_error.restype = ctypes.c_char_p
    for name in ("raz_net_weight_bytes", "raz_net_scratch_bytes", "raz_net_load", "raz_net_forward", "raz_net_range_stats"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = N.SIGNATURES[name]
    return lib


def emu_forward(lib, blob, own, enemy, reserved):
    from reversi_alpha_zero_amd import _native as N
    _, _, F, R, V = struct.unpack_from("<5i", blob, 0)
    w = np.zeros(lib.raz_net_weight_bytes(F, R, V), dtype=np.uint8)
    net = N.RazNet()
    net.reserved = reserved
    if lib.raz_net_load(ctypes.byref(net), blob, len(blob), w.ctypes.data, w.size, None) != 0:
        raise RuntimeError(lib.raz_last_error())
    n = len(own)
    need = lib.raz_net_scratch_bytes(F, V, n)
    scratch = np.zeros(max(need, 8), dtype=np.uint8)
    pol, val = np.zeros((n, 64), np.float32), np.zeros(n, np.float32)
    own, enemy = np.ascontiguousarray(own, dtype=np.uint64), np.ascontiguousarray(enemy, dtype=np.uint64)
    if lib.raz_net_forward(ctypes.byref(net), own.ctypes.data, enemy.ctypes.data, None, pol.ctypes.data, val.ctypes.data, n,
                           scratch.ctypes.data if need else None, need, None) != 0:
        raise RuntimeError(lib.raz_last_error())
    over, rows = ctypes.c_int(0), ctypes.c_ulonglong(0)
    lib.raz_net_range_stats(ctypes.byref(net), ctypes.byref(over), ctypes.byref(rows), None)
    return pol, val, bool(over.value), int(rows.value)


def err(a, b):
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    return {"max": float(d.max()), "mean": float(d.mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=600)
    ap.add_argument("--sims", type=int, default=30)
    ap.add_argument("--steps", type=int, default=2500)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--net", default="256,10,256")
    ap.add_argument("--emu-positions", type=int, default=16, help="held-out positions evaluated by the emulated kernels (a 256x10 forward is ~1 min each there)")
    ap.add_argument("--torch-positions", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--save-blob", default=None, help="write the trained net's raznet blob here (94 MB for 256x10: not for the repository)")
    a = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    F, R, V = (int(x) for x in a.net.split(","))
    torch.manual_seed(0)
    torch.set_num_threads(a.threads)

    t0 = time.time()
    rows = rows_from_oracle_games(a.games, a.sims, 31, a.threads)
    held = rows_from_oracle_games(max(8, a.games // 10), a.sims, 77, a.threads)
    t_data = time.time() - t0
    X = torch.from_numpy(planes_of([r[0] for r in rows], [r[1] for r in rows]))
    P = torch.from_numpy(np.stack([r[2] for r in rows]))
    Z = torch.tensor([r[3] for r in rows], dtype=torch.float32).view(-1, 1)

    net = ReversiNet(F, R, V).keras_init_(0)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.01   # Keras BatchNormalization(momentum=0.99)
    init_blob = net.to_blob()
    opt = torch.optim.SGD(net.parameters(), lr=a.lr, momentum=0.9, weight_decay=1e-4)
    net.train()
    gen = torch.Generator().manual_seed(5)
    losses = []
    t0 = time.time()
    for step in range(a.steps):
        idx = torch.randint(0, X.shape[0], (a.batch,), generator=gen)
        p, v = net(X[idx])
        loss_p = -(P[idx] * torch.log(p.clamp_min(1e-12))).sum(1).mean()
        loss_v = ((v - Z[idx]) ** 2).mean()
        loss = loss_p + loss_v
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append((float(loss_p), float(loss_v)))
        if step % 100 == 0:
            print(f"step {step}: policy {loss_p:.4f} value {loss_v:.4f}  ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    t_train = time.time() - t0
    net.eval()
    blob = net.to_blob()
    if a.save_blob:
        with open(a.save_blob, "wb") as f:
            f.write(blob)

    # what training did to the numbers the kernels see: the folded per-channel scale gamma / sqrt(var + eps) of every conv + BN
    scales, wabs = [], []
    for cb in [net.stem] + [c for blk in net.res for c in blk]:
        s = (cb.bn.weight.detach().double() / torch.sqrt(cb.bn.running_var.detach().double() + cb.bn.eps)).abs()
        scales.append(s)
        wabs.append(float(cb.conv.weight.detach().abs().max()))
    s = torch.cat(scales)
    fold = {"folded_bn_scale_min": float(s.min()), "folded_bn_scale_max": float(s.max()), "decades_between": float(torch.log10(s.max() / s.min())),
            "conv_weight_abs_max": max(wabs),
            "weights_moved_from_init_rel_l2": float(np.linalg.norm(np.frombuffer(blob, np.float32, offset=32) - np.frombuffer(init_blob, np.float32, offset=32))
                                                    / np.linalg.norm(np.frombuffer(init_blob, np.float32, offset=32)))}

    # held-out positions
    ho = np.array([r[0] for r in held], dtype=np.uint64)[:a.torch_positions]
    he = np.array([r[1] for r in held], dtype=np.uint64)[:a.torch_positions]
    hx = torch.from_numpy(planes_of(ho, he))
    with torch.no_grad():
        p32, v32 = net(hx)
        net64 = ReversiNet(F, R, V)
        net64.load_state_dict(net.state_dict())
        net64 = net64.double().eval()
        p64, v64 = net64(hx.double())
    hp = np.stack([r[2] for r in held])[:a.torch_positions]
    hz = np.array([r[3] for r in held], dtype=np.float32)[:a.torch_positions]
    fit = {"held_out_policy_cross_entropy": float(-(hp * np.log(np.clip(p32.numpy(), 1e-12, None))).sum(1).mean()),
           "held_out_value_mse": float(((v32.numpy().reshape(-1) - hz) ** 2).mean()),
           "held_out_policy_entropy_of_targets": float(-(hp * np.log(np.clip(hp, 1e-12, None))).sum(1).mean())}

    lib = emu_lib()
    k = min(a.emu_positions, len(ho))
    pick = np.linspace(0, len(ho) - 1, k).astype(int)
    t0 = time.time()
    pol1, val1, _, _ = emu_forward(lib, blob, ho[pick], he[pick], 0)
    t_v1 = time.time() - t0
    t0 = time.time()
    pol2, val2, over2, rows2 = emu_forward(lib, blob, ho[pick], he[pick], 4)
    t_v2 = time.time() - t0
    pol2i, val2i, _, _ = emu_forward(lib, init_blob, ho[pick[:max(2, k // 4)]], he[pick[:max(2, k // 4)]], 4)
    with torch.no_grad():
        neti = ReversiNet(F, R, V).keras_init_(0)
        pi32, vi32 = neti(hx[pick[:max(2, k // 4)]])
    out = {
        "what": "raznet-forward-v2 (split-f16 trunk) and v1 (exact f32) of the PRODUCT's kernel sources, compiled for the host by the wave emulator, on a net "
                "TRAINED on the host cores - against the fp32 torch graph (the tolerance's reference) and the same graph in f64",
        "net": [F, R, V], "tolerance_of_the_north_star": 1e-5,
        "training": {"rows": len(rows), "from": f"{a.games} oracle self-play games (mini net, {a.sims} sims/move), searched plies only", "steps": a.steps, "batch": a.batch,
                     "recipe": f"SGD lr {a.lr} momentum 0.9, weight decay 1e-4, loss = policy cross-entropy + value MSE, BatchNorm momentum 0.99 (worker/optimize.py:72-111)",
                     "loss_first_50_steps_mean": [float(np.mean([l[i] for l in losses[:50]])) for i in (0, 1)],
                     "loss_last_50_steps_mean": [float(np.mean([l[i] for l in losses[-50:]])) for i in (0, 1)],
                     "seconds": {"data": t_data, "training": t_train}, "host_threads": a.threads, **fit},
        "what_training_changed": fold,
        "fp32_torch_vs_f64_torch": {"positions": int(len(ho)), "policy": err(p32.numpy(), p64.numpy()), "value": err(v32.numpy(), v64.numpy())},
        "emulated_kernels": {
            "positions": int(k), "seconds_per_forward_batch": {"v1": t_v1, "v2": t_v2},
            "v1_vs_fp32_torch": {"policy": err(pol1, p32.numpy()[pick]), "value": err(val1, v32.numpy().reshape(-1)[pick])},
            "v2_vs_fp32_torch": {"policy": err(pol2, p32.numpy()[pick]), "value": err(val2, v32.numpy().reshape(-1)[pick])},
            "v2_vs_f64_torch": {"policy": err(pol2, p64.numpy()[pick]), "value": err(val2, v64.numpy().reshape(-1)[pick])},
            "v1_vs_f64_torch": {"policy": err(pol1, p64.numpy()[pick]), "value": err(val1, v64.numpy().reshape(-1)[pick])},
            "v2_vs_v1": {"policy": err(pol2, pol1), "value": err(val2, val1)},
            "v2_range_flag": over2, "v2_rows_repaired": rows2,
            "same_positions_on_the_untrained_initialisation_v2_vs_fp32_torch": {
                "positions": int(len(pol2i)), "policy": err(pol2i, pi32.numpy()), "value": err(val2i, vi32.numpy().reshape(-1))},
        },
    }
    e = out["emulated_kernels"]
    out["within_tolerance"] = bool(max(e["v2_vs_fp32_torch"]["policy"]["max"], e["v2_vs_fp32_torch"]["value"]["max"]) <= 1e-5)
    text = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
