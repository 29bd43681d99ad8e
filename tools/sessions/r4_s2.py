#!/usr/bin/env python
"""Round 4, GPU session 2: raznet-forward-v3 (k_conv3x3_wino) on the device - accuracy against fp32 torch (and f64 on a sample) next to
raznet-forward-v2 on harvested positions x 2 nets, batch invariance bit for bit, and ms per 8192-position forward for both."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from bench_sweep import harvest_positions
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    dev = torch.device("cuda:0")
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    n = 8192
    black, white, player, _ = harvest_positions(n, 99, dev)
    own = torch.where(player == 1, black, white)
    enemy = torch.where(player == 1, white, black)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    planes = torch.stack([((own[:, None] >> sh) & 1), ((enemy[:, None] >> sh) & 1)], dim=1).float().reshape(-1, 2, 8, 8)
    out = {}
    for name, seed, bn in (("bench net (keras_init 0)", 0, None), ("seed 5, BN stats in [0.5, 1.5]", 5, 6)):
        net = ReversiNet(256, 10, 256).keras_init_(seed)
        if bn is not None:
            net.randomize_bn_(bn)
        net.eval()
        blob = net.to_blob()
        n32 = ReversiNet(256, 10, 256)
        n32.load_state_dict(net.state_dict())
        n32 = n32.to(dev).eval()
        n64 = ReversiNet(256, 10, 256)
        n64.load_state_dict(net.state_dict())
        n64 = n64.double().to(dev).eval()
        with torch.no_grad():
            tp, tv = n32(planes)
            dp, dv = n64(planes[:2048].double())
        row = {"torch_fp32_vs_f64": {"policy_max": float((tp[:2048].double() - dp).abs().max()), "value_max": float((tv[:2048, 0].double() - dv[:, 0]).abs().max())}}
        res = {}
        for k in ("f16x3", "wino"):
            dn = DeviceNet(blob, dev, kernel=k)
            p, v = dn.predict_bitboards(own, enemy)
            torch.cuda.synchronize()
            res[k] = (p.clone(), v.clone())
            row[k] = {"vs_torch_fp32": {"policy_max": float((p - tp).abs().max()), "value_max": float((v - tv[:, 0]).abs().max())},
                      "vs_f64_first_2048": {"policy_max": float((p[:2048].double() - dp).abs().max()), "value_max": float((v[:2048].double() - dv[:, 0]).abs().max())},
                      "range_ok": bool(dn.range_ok())}
            # batch invariance: a ragged sub-batch from the middle gives the same bits
            p2, v2 = dn.predict_bitboards(own[1000:1037].contiguous(), enemy[1000:1037].contiguous())
            row[k]["batch_invariant_bitwise"] = bool(torch.equal(p2.view(torch.int32), p[1000:1037].view(torch.int32)) and torch.equal(v2.view(torch.int32), v[1000:1037].view(torch.int32)))
            if seed == 0:
                for rep in range(2):
                    iters = 10
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
                    ev[0].record()
                    for i in range(iters):
                        dn.predict_bitboards(own, enemy)
                        ev[i + 1].record()
                    torch.cuda.synchronize()
                    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
                    row[k].setdefault("ms_per_forward_8192_median_best", []).append([round(statistics.median(ms), 3), round(min(ms), 3)])
            del dn
        row["wino_vs_f16x3"] = {"policy_max": float((res["wino"][0] - res["f16x3"][0]).abs().max()), "value_max": float((res["wino"][1] - res["f16x3"][1]).abs().max())}
        out[name] = row
        print(json.dumps({name: row}), flush=True)
        del n32, n64
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
