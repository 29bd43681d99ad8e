#!/usr/bin/env python
"""tools/check_net_accuracy.py — error statistics of the device forward kernels against the fp32 torch graph (on the GPU
through PyTorch-ROCm and, for a sample, on the CPU) over many positions harvested from random play:
    python tools/check_net_accuracy.py [--n 2048] [--net ch5]
Prints one JSON line: max / 99.9th percentile / mean absolute error of policy and value for the exact-f32 kernels
(raznet-forward-v1) and the split-f16 kernels (raznet-forward-v2), and v2 vs v1."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--net", default="ch5", choices=["ch5", "w128"])
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from bench_sweep import harvest_positions
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    dev = torch.device("cuda:0")
    F, R, V = {"ch5": (256, 10, 256), "w128": (128, 4, 128)}[a.net]
    net = ReversiNet(F, R, V).keras_init_(a.seed).randomize_bn_(a.seed + 1)
    blob = net.to_blob()
    black, white, player, _ = harvest_positions(a.n, 99, dev)
    own = torch.where(player == 1, black, white)
    enemy = torch.where(player == 1, white, black)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    planes = torch.stack([((own[:, None] >> sh) & 1), ((enemy[:, None] >> sh) & 1)], dim=1).float().reshape(-1, 2, 8, 8)
    with torch.no_grad():
        tp, tv = net.to(dev)(planes)
        tv = tv[:, 0]
        m = min(a.n, 64)
        cp, cv = net.cpu()(planes[:m].cpu())
    out = {"net": [F, R, V], "positions": a.n}
    res = {}
    for k in ("f32", "f16x3"):
        dn = DeviceNet(blob, dev, kernel=k)
        p, v = dn.predict_bitboards(own, enemy)
        res[k] = (p, v)
        dp, dv = (p - tp).abs().flatten(), (v - tv).abs()
        out[k] = {"policy": {"max": float(dp.max()), "p999": float(dp.quantile(0.999)), "mean": float(dp.mean())},
                  "value": {"max": float(dv.max()), "p999": float(dv.quantile(0.999)), "mean": float(dv.mean())},
                  "vs_torch_cpu_first64": {"policy_max": float((p[:m].cpu() - cp).abs().max()), "value_max": float((v[:m].cpu() - cv[:, 0]).abs().max())},
                  "range_ok": dn.range_ok()}
    out["f16x3_vs_f32"] = {"policy_max": float((res["f16x3"][0] - res["f32"][0]).abs().max()),
                           "value_max": float((res["f16x3"][1] - res["f32"][1]).abs().max())}
    out["torch_gpu_vs_cpu_first64"] = {"policy_max": float((tp[:m].cpu() - cp).abs().max()), "value_max": float((tv[:m].cpu() - cv[:, 0]).abs().max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
