#!/usr/bin/env python
"""k_net_mfma16_split (one position on two / four waves) against k_net_mfma on the device, in ONE process: outputs bit for bit
(several shapes, ragged sizes, an active mask), then stand-alone time per forward (median / best of 100) at the sizes the
engine launches.  Prints one JSON line."""
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    from reversi_alpha_zero_amd._native import lib, check
    dev = torch.device("cuda:0")
    out = {}

    def fwd(net, o, e, act):
        n = o.numel()
        pol = torch.full((n, 64), 7.0, dtype=torch.float32, device=dev)
        val = torch.full((n,), 7.0, dtype=torch.float32, device=dev)
        check(lib.raz_net_forward(ctypes.byref(net.c), o.data_ptr(), e.data_ptr(), act.data_ptr() if act is not None else None,
                                  pol.data_ptr(), val.data_ptr(), n, None, 0, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return pol, val

    for shape in ((16, 1, 16), (16, 2, 48), (16, 1, 80)):
        blob = ReversiNet(*shape).keras_init_(3).randomize_bn_(4).to_blob()
        nets = {k: DeviceNet(blob, dev, kernel=k) for k in ("mfma_wave", "mfma_split2", "mfma_split4")}
        ok = True
        for n in (1, 63, 1365, 20000):
            rng = np.random.default_rng(n)
            own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
            enemy = rng.integers(0, 2**64, size=n, dtype=np.uint64) & ~own
            o, e = torch.from_numpy(own.view(np.int64)).to(dev), torch.from_numpy(enemy.view(np.int64)).to(dev)
            act = torch.from_numpy((rng.integers(0, 4, size=n) != 0).astype(np.uint8)).to(dev)
            for a in (None, act):
                ref = fwd(nets["mfma_wave"], o, e, a)
                for k in ("mfma_split2", "mfma_split4"):
                    got = fwd(nets[k], o, e, a)
                    ok = ok and bool(torch.equal(ref[0].view(torch.int32), got[0].view(torch.int32))
                                     and torch.equal(ref[1].view(torch.int32), got[1].view(torch.int32)))
        out["bit_equal_%dx%d_v%d" % shape] = ok
    blob = ReversiNet(16, 1, 16).keras_init_(0).to_blob()
    nets = {k: DeviceNet(blob, dev, kernel=k) for k in ("mfma_wave", "mfma_split2", "mfma_split4")}
    for n in (512, 1024, 1365, 2048, 4096, 16384):
        rng = np.random.default_rng(0)
        own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
        enemy = rng.integers(0, 2**64, size=n, dtype=np.uint64) & ~own
        o, e = torch.from_numpy(own.view(np.int64)).to(dev), torch.from_numpy(enemy.view(np.int64)).to(dev)
        pol = torch.empty((n, 64), dtype=torch.float32, device=dev)
        val = torch.empty(n, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for k, net in nets.items():
            call = lambda: lib.raz_net_forward(ctypes.byref(net.c), o.data_ptr(), e.data_ptr(), None, pol.data_ptr(), val.data_ptr(), n, None, 0, st)
            for _ in range(5):
                call()
            iters = 100
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
            ev[0].record()
            for i in range(iters):
                call()
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
            out[f"{k}_n{n}_us"] = [round(statistics.median(ms) * 1e3, 2), round(min(ms) * 1e3, 2)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
