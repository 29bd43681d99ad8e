#!/usr/bin/env python
"""tools/trained_net.py - the policy/value net TRAINED on the MI355X, then the device forward kernels on it.

Every figure of raznet-forward-v2 (the split-f16 trunk of csrc/raz_net_f16x3.hip - the headline's net kernel) on the GPU was taken on
random initialisations.  This tool runs the reference's loop on one device, small:
  generation g:  self-play of --games games by the engine with the CURRENT 256x10 net (reversi-alpha-zero_amd/engine.py, ch5.yml play
                 settings without resignation, --sims simulations per move) -> rows (own, enemy, root visit distribution, z) exactly as
                 the reference's worker stores them (worker/self_play.py:180-194, agent/player.py:166-179);
                 --steps SGD steps of --batch rows drawn from all rows so far with a random D4 symmetry, the reference's recipe
                 (worker/optimize.py:72-111: SGD momentum 0.9, categorical cross-entropy + mean squared error, l2 1e-4; BatchNorm
                 momentum 0.99 as in Keras) in torch-ROCm fp32.
Generation 0 plays with the random initialisation (what the reference does before its first checkpoint).  After the last generation:
  * >= --positions positions (half from a held-out self-play batch of the TRAINED net, half from on-device random playouts frozen
    at a uniformly random ply) through the exact-f32 kernels (v1), the split-f16 kernels (v2), the fp32 torch graph on the GPU - the
    tolerance's reference - and the SAME graph in f64 (an explicit im2col + f64 GEMM restatement, checked here against torch's own
    f64 convolution on a sample: MIOpen's f64 convolutions are too slow for 65 536 positions);
  * one JSON document: per pair max / 99.9th percentile / mean absolute error of policy and value, what training did to the folded
    BatchNorm scales, the fit on held-out rows, and the two bounds tests/test_net_trained_gpu.py asserts on a shorter run of the same
    recipe: err(v2, f64) <= 1.5 x err(torch fp32, f64) and err(v2, f64) <= 1e-5.
    python tools/trained_net.py --out gpurun_out/net_v2_on_a_gpu_trained_256x10_net.json      (about 4 minutes on an MI355X)"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def play_config(sims):
    """config/ch5.yml:9-16 over config.py:128-166, one simulation in flight, thinking_loop 1, solver off, NO resignation (whole games:
    every ply of every game is a training row)."""
    play = types.SimpleNamespace(
        simulation_num_per_move=sims, share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=400,
        start_rethinking_turn=8, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=4, virtual_loss=3, parallel_search_num=1,
        resign_threshold=None, allowed_resign_turn=50, disable_resignation_rate=0.1, use_solver_turn=0, use_solver_turn_in_simulation=0)
    return types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))


def selfplay_rows(blob, dev, n_games, sims, seed, first_id, kernel="auto"):
    """Whole self-play games of the engine -> device tensors (own i64 [n], enemy i64 [n], policy f32 [n, 64], z f32 [n]) of every
    searched ply, mover's view; plus {"games", "rows", "seconds", "black_wins", "white_wins", "net_kernel", "range_ok"}."""
    import torch
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    net = DeviceNet(blob, dev, kernel=kernel)
    eng = SelfPlayEngine(play_config(sims), net, n_games=n_games, seed=seed, sims_hint=sims)
    t0 = time.time()
    eng.start(first_id, sims)
    eng.run(chunk=64)
    raw = eng.read_raw()
    dt = time.time() - t0
    hdr, rn, npl, status = raw["headers"], raw["root_n"], raw["n_plies"], raw["status"]
    winner = status & 0x0f
    black_win = np.where(winner == 1, 1.0, np.where(winner == 2, -1.0, 0.0)).astype(np.float32)
    ply = np.arange(hdr.shape[1])[None, :]
    tot = rn.sum(axis=2, dtype=np.int64)
    keep = (ply < npl[:, None]) & (hdr["has_row"] != 0) & (tot > 0)
    g, p = np.nonzero(keep)
    own = hdr["own"][g, p].astype(np.uint64).view(np.int64)
    enemy = hdr["enemy"][g, p].astype(np.uint64).view(np.int64)
    pol = (rn[g, p].astype(np.float64) / tot[g, p][:, None]).astype(np.float32)
    z = np.where(hdr["player"][g, p] == 1, black_win[g], -black_win[g]).astype(np.float32)
    info = {"games": int(n_games), "rows": int(len(g)), "seconds": dt, "black_wins": int((winner == 1).sum()), "white_wins": int((winner == 2).sum()),
            "net_kernel": net.kernel_name, "range_ok": bool(net.range_ok()), "sims_per_move": int(sims)}
    del eng, net
    torch.cuda.empty_cache()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t(own), t(enemy), t(pol), t(z), info


def planes_of(own, enemy):
    """int64 device bitboards -> [n, 2, 8, 8] f32 planes [own, enemy] (bit i = square i, lib/bitboard.py:10-17)."""
    import torch
    sh = torch.arange(64, device=own.device, dtype=torch.int64)
    return torch.stack([((own[:, None] >> sh) & 1), ((enemy[:, None] >> sh) & 1)], dim=1).float().reshape(-1, 2, 8, 8)


def d4(x, k, flip):
    """One of the 8 symmetries of the square on the last two axes (the rows agent/player.py:166-179 stores come in all eight)."""
    import torch
    x = torch.rot90(x, k, dims=(-2, -1))
    return torch.flip(x, dims=(-1,)) if flip else x


def train(net, rows, steps, batch, lr, seed, log=None):
    """The reference's recipe on rows = (own, enemy, policy, z) device tensors; returns the per-step (policy, value) losses."""
    import torch
    own, enemy, pol, z = rows
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.01   # Keras BatchNormalization(momentum=0.99)
    opt = torch.optim.SGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    net.train()
    gen = torch.Generator(device=own.device).manual_seed(seed)
    cpu_gen = np.random.default_rng(seed)
    losses = []
    for step in range(steps):
        idx = torch.randint(0, own.numel(), (batch,), generator=gen, device=own.device)
        k, flip = int(cpu_gen.integers(4)), bool(cpu_gen.integers(2))
        x = d4(planes_of(own[idx], enemy[idx]), k, flip)
        target = d4(pol[idx].reshape(-1, 8, 8), k, flip).reshape(-1, 64)
        p, v = net(x)
        loss_p = -(target * torch.log(p.clamp_min(1e-12))).sum(1).mean()
        loss_v = ((v[:, 0] - z[idx]) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        (loss_p + loss_v).backward()
        opt.step()
        if step % 50 == 0 or step == steps - 1:
            losses.append((step, float(loss_p.detach()), float(loss_v.detach())))
            if log:
                log(f"  step {step}: policy {losses[-1][1]:.4f} value {losses[-1][2]:.4f}")
    net.eval()
    return losses


class F64Graph:
    """agent/model.py:28-72 in f64 as explicit GEMMs: conv3x3 "same" = im2col (nine shifted views) x [F, C*9] matrix, BatchNorm applied
    as the f64 affine map it is at inference (eps 1e-3), heads as in ReversiNet.forward.  The answer every fp32 evaluation is measured
    against."""

    def __init__(self, net, dev):
        import torch
        d = lambda t: t.detach().to(dev).double()

        def fold(cb):
            s = d(cb.bn.weight) / torch.sqrt(d(cb.bn.running_var) + cb.bn.eps)
            w = d(cb.conv.weight)
            return (w.reshape(w.shape[0], -1) * s[:, None]).contiguous(), ((d(cb.conv.bias) - d(cb.bn.running_mean)) * s + d(cb.bn.bias))
        self.stem = fold(net.stem)
        self.res = [(fold(c1), fold(c2)) for c1, c2 in net.res]
        self.pc, self.vc = fold(net.policy_conv), fold(net.value_conv)
        self.pfc = (d(net.policy_fc.weight), d(net.policy_fc.bias))
        self.vf1 = (d(net.value_fc1.weight), d(net.value_fc1.bias))
        self.vf2 = (d(net.value_fc2.weight), d(net.value_fc2.bias))

    @staticmethod
    def conv3(x, wb):
        import torch
        import torch.nn.functional as Fn
        n, c = x.shape[:2]
        xp = Fn.pad(x, (1, 1, 1, 1))
        cols = torch.stack([xp[:, :, dy:dy + 8, dx:dx + 8] for dy in range(3) for dx in range(3)], dim=2).reshape(n, c * 9, 64)
        return (torch.matmul(wb[0], cols) + wb[1][None, :, None]).reshape(n, -1, 8, 8)

    @staticmethod
    def conv1(x, wb):
        import torch
        n = x.shape[0]
        return (torch.matmul(wb[0], x.reshape(n, x.shape[1], 64)) + wb[1][None, :, None]).reshape(n, -1, 8, 8)

    def __call__(self, planes, chunk=1024):
        import torch
        ps, vs = [], []
        with torch.no_grad():
            for c0 in range(0, planes.shape[0], chunk):
                x = torch.relu(self.conv3(planes[c0:c0 + chunk].double(), self.stem))
                for c1, c2 in self.res:
                    x = torch.relu(self.conv3(torch.relu(self.conv3(x, c1)), c2) + x)
                p = torch.relu(self.conv1(x, self.pc)).flatten(1)
                ps.append(torch.softmax(p @ self.pfc[0].t() + self.pfc[1], dim=1))
                v = torch.relu(self.conv1(x, self.vc)).flatten(1)
                v = torch.relu(v @ self.vf1[0].t() + self.vf1[1])
                vs.append(torch.tanh(v @ self.vf2[0].t() + self.vf2[1])[:, 0])
        return torch.cat(ps), torch.cat(vs)


def err_stats(a, b):
    d = (a.double() - b.double()).abs().flatten()
    k = max(1, int(round(d.numel() * 0.999)))
    return {"max": float(d.max()), "p999": float(d.kthvalue(k).values), "mean": float(d.mean())}


def evaluate(net, blob, own, enemy, dev, chunk=8192, miopen_f64_sample=256):
    """v1 / v2 kernels, fp32 torch (GPU) and the f64 graph on the given positions -> the error table."""
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    n32 = net.to(dev).eval()
    g64 = F64Graph(net, dev)
    nets = {"v1_exact_f32": DeviceNet(blob, dev, kernel="f32"), "v2_split_f16": DeviceNet(blob, dev, kernel="f16x3")}
    outs = {k: ([], []) for k in ("v1_exact_f32", "v2_split_f16", "torch_fp32", "f64")}
    t = {k: 0.0 for k in outs}
    for c0 in range(0, own.numel(), chunk):
        o, e = own[c0:c0 + chunk].contiguous(), enemy[c0:c0 + chunk].contiguous()
        planes = planes_of(o, e)
        for k, dn in nets.items():
            torch.cuda.synchronize()
            t0 = time.time()
            p, v = dn.predict_bitboards(o, e)
            torch.cuda.synchronize()
            t[k] += time.time() - t0
            outs[k][0].append(p)
            outs[k][1].append(v)
        with torch.no_grad():
            t0 = time.time()
            p, v = n32(planes)
            torch.cuda.synchronize()
            t["torch_fp32"] += time.time() - t0
        outs["torch_fp32"][0].append(p)
        outs["torch_fp32"][1].append(v[:, 0])
        t0 = time.time()
        p, v = g64(planes)
        torch.cuda.synchronize()
        t["f64"] += time.time() - t0
        outs["f64"][0].append(p)
        outs["f64"][1].append(v)
    res = {k: (torch.cat(a), torch.cat(b)) for k, (a, b) in outs.items()}
    table = {}
    for k in ("v1_exact_f32", "v2_split_f16", "torch_fp32"):
        table[f"{k}_vs_f64"] = {"policy": err_stats(res[k][0], res["f64"][0]), "value": err_stats(res[k][1], res["f64"][1])}
    for k in ("v1_exact_f32", "v2_split_f16"):
        table[f"{k}_vs_torch_fp32"] = {"policy": err_stats(res[k][0], res["torch_fp32"][0]), "value": err_stats(res[k][1], res["torch_fp32"][1])}
    table["v2_vs_v1"] = {"policy": err_stats(res["v2_split_f16"][0], res["v1_exact_f32"][0]), "value": err_stats(res["v2_split_f16"][1], res["v1_exact_f32"][1])}
    ok, repaired = nets["v2_split_f16"].range_stats()
    table["v2_range_flag_clear"], table["v2_rows_repaired_on_the_exact_chains"] = bool(ok), int(repaired)
    # the f64 restatement against torch's own f64 graph (MIOpen / rocBLAS f64) on a sample
    m = min(miopen_f64_sample, own.numel())
    n64 = ReversiNet(net.filters, net.res_layers, net.value_fc)
    n64.load_state_dict(net.state_dict())
    n64 = n64.double().to(dev).eval()
    with torch.no_grad():
        p64, v64 = n64(planes_of(own[:m].contiguous(), enemy[:m].contiguous()).double())
    table["f64_gemm_restatement_vs_torch_f64_module"] = {"positions": int(m), "policy_max": float((p64 - res["f64"][0][:m]).abs().max()),
                                                         "value_max": float((v64[:, 0] - res["f64"][1][:m]).abs().max())}
    table["seconds"] = t
    table["positions"] = int(own.numel())
    vmax = lambda key: max(table[key]["policy"]["max"], table[key]["value"]["max"])
    table["bounds"] = {
        "err_v2_vs_f64_max": vmax("v2_split_f16_vs_f64"), "err_torch_fp32_vs_f64_max": vmax("torch_fp32_vs_f64"), "err_v1_vs_f64_max": vmax("v1_exact_f32_vs_f64"),
        "v2_within_1e-5_of_f64": bool(vmax("v2_split_f16_vs_f64") <= 1e-5),
        "v2_within_1.5x_of_what_fp32_torch_itself_is_from_f64": bool(vmax("v2_split_f16_vs_f64") <= 1.5 * vmax("torch_fp32_vs_f64")),
        "v2_within_1e-5_of_torch_fp32": bool(vmax("v2_split_f16_vs_torch_fp32") <= 1e-5),
        "v1_within_1e-5_of_torch_fp32": bool(vmax("v1_exact_f32_vs_torch_fp32") <= 1e-5)}
    del nets
    torch.cuda.empty_cache()
    return table


def what_training_changed(net, init_blob, blob):
    import torch
    scales, wabs = [], []
    for cb in [net.stem] + [c for blk in net.res for c in blk]:
        scales.append((cb.bn.weight.detach().double() / torch.sqrt(cb.bn.running_var.detach().double() + cb.bn.eps)).abs().cpu())
        wabs.append(float(cb.conv.weight.detach().abs().max()))
    s = torch.cat(scales)
    a, b = np.frombuffer(blob, np.float32, offset=32), np.frombuffer(init_blob, np.float32, offset=32)
    return {"folded_bn_scale_min": float(s.min()), "folded_bn_scale_max": float(s.max()), "decades_between": float(torch.log10(s.max() / s.min())),
            "conv_weight_abs_max": max(wabs), "weights_moved_from_init_rel_l2": float(np.linalg.norm(a - b) / np.linalg.norm(b))}


def train_generations(dev, shape, generations, games, sims, steps, batch, lr, seed, log=None):
    """The loop of the docstring.  Returns (net on `dev` in eval mode, init blob, blob, rows of all generations, per-generation report)."""
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    torch.manual_seed(seed)
    F, R, V = shape
    net = ReversiNet(F, R, V).keras_init_(seed)
    init_blob = blob = net.to_blob()
    net = net.to(dev)
    rows, report, first_id = None, [], 0
    for g in range(generations):
        own, enemy, pol, z, info = selfplay_rows(blob, dev, games, sims, seed, first_id)
        first_id += games
        rows = (own, enemy, pol, z) if rows is None else tuple(torch.cat([a, b]) for a, b in zip(rows, (own, enemy, pol, z)))
        if log:
            log(f"generation {g}: {info}")
        t0 = time.time()
        losses = train(net, rows, steps, batch, lr, seed + 17 * g, log)
        blob = net.cpu().to_blob()
        net = net.to(dev)
        report.append({"generation": g, "self_play": info, "rows_in_the_window": int(rows[0].numel()), "sgd_steps": steps, "training_seconds": time.time() - t0,
                       "loss_first": losses[0][1:], "loss_last": losses[-1][1:]})
    return net, init_blob, blob, rows, report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="256,10,256")
    ap.add_argument("--generations", type=int, default=3)
    ap.add_argument("--games", type=int, default=2048)
    ap.add_argument("--sims", type=int, default=40)
    ap.add_argument("--steps", type=int, default=1500, help="SGD steps per generation")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--positions", type=int, default=65536)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from bench_sweep import harvest_positions
    dev = torch.device("cuda:0")
    log = lambda s: print(s, file=sys.stderr, flush=True)
    shape = tuple(int(x) for x in a.net.split(","))
    t_all = time.time()
    net, init_blob, blob, rows, report = train_generations(dev, shape, a.generations, a.games, a.sims, a.steps, a.batch, a.lr, a.seed, log)
    # held-out positions: a fresh self-play batch of the TRAINED net (ids never trained on) + random playouts
    half = a.positions // 2
    h_games = max(64, (half + 57) // 58 + 32)
    ho, he, hp, hz, hinfo = selfplay_rows(blob, dev, h_games, a.sims, a.seed, 10_000_000)
    log(f"held-out self-play of the trained net: {hinfo}")
    black, white, player, _ = harvest_positions(a.positions - min(half, ho.numel()), 99, dev)
    ro, re = torch.where(player == 1, black, white), torch.where(player == 1, white, black)
    own = torch.cat([ho[:half], ro]).contiguous()
    enemy = torch.cat([he[:half], re]).contiguous()
    with torch.no_grad():
        p32, v32 = net(planes_of(ho[:8192].contiguous(), he[:8192].contiguous()))
    fit = {"held_out_rows": int(min(8192, ho.numel())),
           "policy_cross_entropy": float(-(hp[:8192] * torch.log(p32.clamp_min(1e-12))).sum(1).mean()),
           "entropy_of_the_targets": float(-(hp[:8192] * torch.log(hp[:8192].clamp_min(1e-12))).sum(1).mean()),
           "value_mse": float(((v32[:, 0] - hz[:8192]) ** 2).mean()), "value_mse_of_predicting_0": float((hz[:8192] ** 2).mean())}
    table = evaluate(net.cpu(), blob, own, enemy, dev)
    # the same positions through the same kernels on the INITIALISATION (what every earlier GPU figure was taken on)
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    init_net = ReversiNet(*shape).keras_init_(a.seed)
    table_init = evaluate(init_net, init_blob, own[:16384].contiguous(), enemy[:16384].contiguous(), dev)
    out = {
        "what": "raznet-forward-v2 (split-f16 trunk, the headline's kernel) and v1 (exact f32) on a net TRAINED ON THIS GPU by the reference's loop "
                "(self-play with the current net -> SGD), against the fp32 torch graph on the GPU and the same graph in f64",
        "device": torch.cuda.get_device_name(0), "net": list(shape), "tolerance_of_the_north_star": 1e-5,
        "loop": report, "what_training_changed": what_training_changed(net.cpu(), init_blob, blob), "fit_on_held_out_self_play_rows": fit,
        "positions": {"total": int(own.numel()), "from_held_out_self_play_of_the_trained_net": int(min(half, ho.numel())),
                      "from_random_playouts_frozen_at_a_uniform_ply": int(ro.numel()), "held_out_self_play": hinfo},
        "trained_net": table, "same_kernels_on_the_initialisation_first_16384_positions": table_init,
        "blob_sha256": __import__("hashlib").sha256(blob).hexdigest(), "seconds_total": time.time() - t_all,
        "command": "python tools/trained_net.py " + " ".join(sys.argv[1:]),
    }
    text = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(json.dumps({"bounds": table["bounds"], "out": a.out}))


if __name__ == "__main__":
    main()
