#!/usr/bin/env python
"""tools/bench_net.py — MFMA roofline of the net forward alone (the "conv batch" leg of the north
star): one raz_net_forward over n positions, repeated; prints one JSON line.
    python tools/bench_net.py [--net ch5|mini] [--n 8192] [--iters 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="ch5", choices=["mini", "ch5"])
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--kernel", default=None, choices=["valu", "mfma_wave", "f32", "f16x3", "auto"],
                    help="force a kernel variant (default: the exact-f32 kernel chosen by shape and n)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from reversi_alpha_zero_amd.agent.model import ReversiNet, macs_per_position
    from reversi_alpha_zero_amd.engine import DeviceNet
    F, R, V = {"mini": (16, 1, 16), "ch5": (256, 10, 256)}[args.net]
    dev = torch.device("cuda:0")
    net = DeviceNet(ReversiNet(F, R, V).keras_init_(0).to_blob(), dev, kernel=args.kernel)
    rng = np.random.default_rng(0)
    own = rng.integers(0, 2**64, size=args.n, dtype=np.uint64)
    enemy = rng.integers(0, 2**64, size=args.n, dtype=np.uint64) & ~own
    o, e = torch.from_numpy(own.view(np.int64)).to(dev), torch.from_numpy(enemy.view(np.int64)).to(dev)
    net.predict_bitboards(o, e)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
    ev[0].record()
    for i in range(args.iters):
        net.predict_bitboards(o, e)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters)]
    flops = 2.0 * macs_per_position(F, R, V) * args.n
    best = min(ms)
    extra = {}
    if os.environ.get("RAZ_NET_PROF") and args.net == "mini":
        import ctypes
        from reversi_alpha_zero_amd._native import lib, check
        prof = torch.zeros(args.n * 8, dtype=torch.int64, device=dev)
        pol = torch.empty((args.n, 64), dtype=torch.float32, device=dev)
        val = torch.empty(args.n, dtype=torch.float32, device=dev)
        check(lib.raz_net_forward(ctypes.byref(net.c), o.data_ptr(), e.data_ptr(), None, pol.data_ptr(), val.data_ptr(),
                                  args.n, prof.data_ptr(), prof.numel() * 8, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        t = prof.cpu().numpy().reshape(args.n, 8).astype(np.float64)
        d = np.diff(t, axis=1)
        names = ["zero_lds", "layer0", "res_blocks", "head_convs", "policy_dense", "softmax", "value_head"]
        extra["phase_ticks_mean"] = {k: float(v) for k, v in zip(names, d.mean(axis=0))}
        extra["wave_ticks_mean"] = float((t[:, 7] - t[:, 0]).mean())
    print(json.dumps({"net": args.net, "kernel": args.kernel or "auto", **extra, "filters": F, "res_layers": R, "positions": args.n, "ms_per_forward": ms,
                      "tflops_best": flops / (best * 1e-3) / 1e12, "peak_tflops_fp32_mfma": 157.3,
                      "frac_of_peak": flops / (best * 1e-3) / 1e12 / 157.3, "flop_per_forward": flops}))


if __name__ == "__main__":
    main()
