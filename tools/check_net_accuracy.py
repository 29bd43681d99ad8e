#!/usr/bin/env python
"""tools/check_net_accuracy.py — error statistics of the device forward kernels over MANY positions harvested from random
play and SEVERAL weight / BatchNorm-statistics variants (random initialisations at several seeds, BN gammas and variances
spread over decades the way a trained checkpoint can be), against the fp32 torch graph on the GPU (PyTorch-ROCm), against
the same graph in f64 (the exact answer up to 1e-16: separates a kernel's own error from the fp32 reference's), and - for a
sample - against fp32 torch on the CPU:
    python tools/check_net_accuracy.py [--n 131072] [--net ch5] [--out profiles/r3/net_accuracy.json]
One JSON document: per variant max / 99.9th percentile / mean absolute error of policy and value for the exact-f32 kernels
(raznet-forward-v1) and the split-f16 kernels (raznet-forward-v2), v2 vs v1, and whether v2's range flag was raised (a variant
that overflows the f16 range is REPORTED as such: the worker replays such a block on the exact-f32 kernels)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

VARIANTS = [  # (name, init seed, BN seed or None, decades)
    ("keras_init seed 0 (the bench net)", 0, None, 0.0),
    ("seed 5, BN stats in [0.5, 1.5]", 5, 6, 0.0),
    ("seed 11, BN gamma/var over 10^+-0.5", 11, 12, 0.5),
    ("seed 17, BN gamma/var over 10^+-1", 17, 18, 1.0),
    ("seed 23, BN gamma/var over 10^+-1.5", 23, 24, 1.5),
]


def stats(d):
    import torch
    d = d.flatten().double()
    k = max(1, int(round(d.numel() * 0.999)))
    return {"max": float(d.max()), "p999": float(d.kthvalue(k).values), "mean": float(d.mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--n64", type=int, default=16384, help="positions also evaluated in f64 (MIOpen's f64 convolutions are slow)")
    ap.add_argument("--net", default="ch5", choices=["ch5", "w128"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import __graft_entry__ as g
    g.build()
    from bench_sweep import harvest_positions
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    dev = torch.device("cuda:0")
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    F, R, V = {"ch5": (256, 10, 256), "w128": (128, 4, 128)}[a.net]
    black, white, player, _ = harvest_positions(a.n, 99, dev)
    own = torch.where(player == 1, black, white)
    enemy = torch.where(player == 1, white, black)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    out = {"net": [F, R, V], "positions": a.n, "positions_from": "on-device random playouts frozen at uniformly random ply (tools/bench_sweep.harvest_positions)",
           "tolerance_of_the_north_star": 1e-5, "variants": []}
    for name, seed, bn_seed, decades in VARIANTS:
        net = ReversiNet(F, R, V).keras_init_(seed)
        if bn_seed is not None:
            net.randomize_bn_(bn_seed, decades=decades)
        net.eval()
        blob = net.to_blob()
        n32 = ReversiNet(F, R, V)
        n32.load_state_dict(net.state_dict())
        n32 = n32.to(dev).eval()
        n64 = ReversiNet(F, R, V)
        n64.load_state_dict(net.state_dict())
        n64 = n64.double().to(dev).eval()
        nets = {k: DeviceNet(blob, dev, kernel=k) for k in ("f32", "f16x3")}
        errs = {k: {r: {"policy": [], "value": []} for r in ("vs_torch_fp32", "vs_torch_f64")} for k in nets}
        errs["torch_fp32"] = {"vs_torch_f64": {"policy": [], "value": []}}
        v21 = {"policy": [], "value": []}
        cpu_cmp = None
        for c0 in range(0, a.n, a.chunk):
            o, e = own[c0:c0 + a.chunk], enemy[c0:c0 + a.chunk]
            planes = torch.stack([((o[:, None] >> sh) & 1), ((e[:, None] >> sh) & 1)], dim=1).float().reshape(-1, 2, 8, 8)
            with_f64 = c0 < a.n64
            with torch.no_grad():
                tp, tv = n32(planes)
                if with_f64:
                    dp, dv = n64(planes.double())
                    dv = dv[:, 0]
            tv = tv[:, 0]
            if with_f64:
                errs["torch_fp32"]["vs_torch_f64"]["policy"].append((tp.double() - dp).abs())
                errs["torch_fp32"]["vs_torch_f64"]["value"].append((tv.double() - dv).abs())
            got = {}
            for k, dn in nets.items():
                p, v = dn.predict_bitboards(o, e)
                got[k] = (p, v)
                errs[k]["vs_torch_fp32"]["policy"].append((p - tp).abs())
                errs[k]["vs_torch_fp32"]["value"].append((v - tv).abs())
                if with_f64:
                    errs[k]["vs_torch_f64"]["policy"].append((p.double() - dp).abs())
                    errs[k]["vs_torch_f64"]["value"].append((v.double() - dv).abs())
            v21["policy"].append((got["f16x3"][0] - got["f32"][0]).abs())
            v21["value"].append((got["f16x3"][1] - got["f32"][1]).abs())
            if c0 == 0:
                m = min(64, o.numel())
                with torch.no_grad():
                    cp, cv = net(planes[:m].cpu())
                cpu_cmp = {k: {"policy_max": float((got[k][0][:m].cpu() - cp).abs().max()), "value_max": float((got[k][1][:m].cpu() - cv[:, 0]).abs().max())}
                           for k in got}
                cpu_cmp["torch_rocm"] = {"policy_max": float((tp[:m].cpu() - cp).abs().max()), "value_max": float((tv[:m].cpu() - cv[:, 0]).abs().max())}
        row = {"variant": name, "f16x3_range_ok": nets["f16x3"].range_ok(), "positions_vs_f64": min(a.n, a.n64),
               "vs_torch_cpu_fp32_first_64_positions": cpu_cmp}
        for k in ("f32", "f16x3", "torch_fp32"):
            row[k] = {r: {h: stats(torch.cat([t.flatten() for t in errs[k][r][h]])) for h in ("policy", "value")} for r in errs[k]}
        row["f16x3_vs_f32"] = {h: stats(torch.cat([t.flatten() for t in v21[h]])) for h in ("policy", "value")}
        row["f16x3_within_1e-5_of_torch_fp32"] = bool(row["f16x3"]["vs_torch_fp32"]["policy"]["max"] <= 1e-5 and row["f16x3"]["vs_torch_fp32"]["value"]["max"] <= 1e-5)
        out["variants"].append(row)
        print(json.dumps({"variant": name, "range_ok": row["f16x3_range_ok"], "f16x3_vs_fp32_policy_max": row["f16x3"]["vs_torch_fp32"]["policy"]["max"],
                          "f16x3_vs_fp32_value_max": row["f16x3"]["vs_torch_fp32"]["value"]["max"],
                          "f32_vs_fp32_policy_max": row["f32"]["vs_torch_fp32"]["policy"]["max"]}), file=sys.stderr, flush=True)
        del n32, n64, nets
        torch.cuda.empty_cache()
    text = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
