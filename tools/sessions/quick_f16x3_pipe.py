#!/usr/bin/env python
"""A/B of the headline conv kernel's hand-scheduled variants (k_conv3x3_f16x3_pipe, RAZ_F16X3_PIPE = 1 .. 4 - read per launch)
in ONE process: outputs of a 256x10 forward over 8192 positions must equal the default kernel's bit for bit; ms per forward
(median / best of `iters`, two rounds) for all five.  Prints one JSON line."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(n=8192, iters=12):
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    dev = torch.device("cuda:0")
    net = DeviceNet(ReversiNet(256, 10, 256).keras_init_(0).to_blob(), dev, kernel="f16x3")
    rng = np.random.default_rng(0)
    own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    enemy = rng.integers(0, 2**64, size=n, dtype=np.uint64) & ~own
    o, e = torch.from_numpy(own.view(np.int64)).to(dev), torch.from_numpy(enemy.view(np.int64)).to(dev)
    out, res = {}, {}
    names = {"0": "default", "1": "pipe_8_waves_x_1_position", "2": "pipe_4_waves_x_2_positions",
             "3": "persistent_pipe_8_waves_x_1_position", "4": "persistent_pipe_4_waves_x_2_positions"}
    order = ("0", "1", "2", "3", "4")
    for mode in order + order:
        os.environ.pop("RAZ_F16X3_PIPE", None)
        if mode != "0":
            os.environ["RAZ_F16X3_PIPE"] = mode
        p, v = net.predict_bitboards(o, e)
        torch.cuda.synchronize()
        res[mode] = (p.clone(), v.clone())
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record()
        for i in range(iters):
            net.predict_bitboards(o, e)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
        out.setdefault(names[mode] + "_ms_per_forward", []).append([round(statistics.median(ms), 3), round(min(ms), 3)])
    os.environ.pop("RAZ_F16X3_PIPE", None)
    same = lambda a, b: bool(torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)))
    out["bit_equal"] = {names[m]: same(res["0"], res[m]) for m in order[1:]}
    out["range_ok"] = bool(net.range_ok())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
