#!/usr/bin/env python
"""tools/eval_winograd_numerics.py — CPU study (no GPU): what would Winograd's minimal filtering cost the split-f16 trunk
(raznet-forward-v2, csrc/raz_net_f16x3.hip) in ACCURACY?  Three arithmetic models of the 256x10 trunk are evaluated on positions
from random play and compared, output by output, with the same graph in f64 (torch CPU) and with fp32 torch:

  direct    the shipped kernel's arithmetic: operands split into (hi, lo) halfs, per (16-channel chunk, tap) three matrix
            instructions al*bh, ah*bl, ah*bh into ONE f32 accumulator (K = 2304 = 144 instruction triples per output);
  wino1d    F(2,3) along x, rows direct: 4 transformed filter points U_a[oc][ic][dy], 4 transformed inputs V_a[ic][row][t]
            (computed in f32 from the layer's f32 result, THEN split), accumulators M_a over (chunk, dy) = 48 triples,
            y[2t] = (M0 + M1) + M2, y[2t+1] = (M1 - M2) - M3 in f32: 1.5x fewer matrix instructions;
  wino2d    F(2x2,3x3): 16 points, K = 256 = 16 triples per accumulator, 2.25x fewer matrix instructions.

Matrix-instruction model: the 16 products of one instruction are summed exactly and added to the f32 accumulator with ONE
rounding (the hardware's internal order is undocumented - tools/probe_f16.hip; this model is what tests/native/wave_emu uses up
to the order).  The heads are the torch fp32 heads for every model, so the differences are the trunk's.
    python tools/eval_winograd_numerics.py [--n 192] [--variant 1] [--out profiles/r4/winograd_numerics.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

VARIANTS = [  # as tools/check_net_accuracy.py
    ("keras_init seed 0 (the bench net)", 0, None, 0.0),
    ("seed 5, BN stats in [0.5, 1.5]", 5, 6, 0.0),
    ("seed 11, BN gamma/var over 10^+-0.5", 11, 12, 0.5),
]


def split(x64):
    """f64 tensor (values representable in f32) -> (hi, lo) as f64 tensors holding f16 values."""
    hi = x64.to(torch.float16)
    lo = (x64 - hi.double()).to(torch.float16)
    return hi.double(), lo.double()


def mfma3(acc, ah, al, bh, bl):
    """acc (f32) += A*B with the three-product split, one f32 rounding per matrix instruction (16-deep k)."""
    for a, b in ((al, bh), (ah, bl), (ah, bh)):
        acc = (acc.double() + a @ b).float()
    return acc


def pow2_scale(w):
    mx = float(w.abs().max())
    e = int(np.frexp(mx)[1]) if mx > 0 else 0
    return 2.0 ** (15 - e)


def conv_direct(x, w, b, skip):
    """x (N,F,8,8) f32 (22-bit values), w (F,F,3,3) f32, b (F).  Returns relu(conv + b (+ skip)) f32."""
    N, F = x.shape[0], x.shape[1]
    S = pow2_scale(w)
    wh, wl = split(w.double() * S)
    xp = torch.zeros((N, F, 10, 10), dtype=torch.float64)
    xp[:, :, 1:9, 1:9] = x.double()
    xh, xl = split(xp)
    acc = torch.zeros((F, N * 64), dtype=torch.float32)
    for c in range(F // 16):
        for t in range(9):
            dy, dx = t // 3, t % 3
            sl = (slice(None), slice(c * 16, c * 16 + 16), slice(dy, dy + 8), slice(dx, dx + 8))
            bh = xh[sl].permute(1, 0, 2, 3).reshape(16, N * 64)
            bl = xl[sl].permute(1, 0, 2, 3).reshape(16, N * 64)
            acc = mfma3(acc, wh[:, c * 16:c * 16 + 16, dy, dx], wl[:, c * 16:c * 16 + 16, dy, dx], bh, bl)
    y = acc * np.float32(1.0 / S)
    return finish(y.reshape(F, N, 8, 8).permute(1, 0, 2, 3), b, skip)


def finish(y, b, skip):
    y = y + b.view(1, -1, 1, 1)
    if skip is not None:
        y = y + skip
    y = torch.relu(y)
    hi, lo = split(y.double())     # what is stored: 22 bits
    return (hi + lo).float()


def conv_wino1d(x, w, b, skip):
    N, F = x.shape[0], x.shape[1]
    g = w.double()                                   # (oc, ic, dy, dx)
    U = torch.stack([g[..., 0], (g[..., 0] + g[..., 1] + g[..., 2]) / 2, (g[..., 0] - g[..., 1] + g[..., 2]) / 2, g[..., 2]]).float().double()
    S = pow2_scale(U)
    Uh, Ul = split(U * S)                            # (4, oc, ic, dy)
    xp = torch.zeros((N, F, 10, 10), dtype=torch.float32)
    xp[:, :, 1:9, 1:9] = x
    d = [xp[:, :, :, k:k + 7:2] for k in range(4)]   # d_k[.., row, t] = xp[.., row, 2t + k]  (x index 2t - 1 + k)
    V = torch.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])   # f32 arithmetic, (4, N, F, 10, 4)
    Vh, Vl = split(V.double())
    M = []
    for a in range(4):
        acc = torch.zeros((F, N * 32), dtype=torch.float32)
        for c in range(F // 16):
            for dy in range(3):
                bh = Vh[a][:, c * 16:c * 16 + 16, dy:dy + 8, :].permute(1, 0, 2, 3).reshape(16, N * 32)
                bl = Vl[a][:, c * 16:c * 16 + 16, dy:dy + 8, :].permute(1, 0, 2, 3).reshape(16, N * 32)
                acc = mfma3(acc, Uh[a][:, c * 16:c * 16 + 16, dy], Ul[a][:, c * 16:c * 16 + 16, dy], bh, bl)
        M.append(acc)
    y0 = (M[0] + M[1]) + M[2]
    y1 = (M[1] - M[2]) - M[3]
    y = torch.stack([y0, y1], dim=-1).reshape(F, N, 8, 4, 2).reshape(F, N, 8, 8) * np.float32(1.0 / S)
    return finish(y.permute(1, 0, 2, 3), b, skip)


def conv_wino2d(x, w, b, skip):
    N, F = x.shape[0], x.shape[1]
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum("ay,oiyx,bx->aboi", G, w.double(), G).float().double()     # (4,4,oc,ic)
    S = pow2_scale(U)
    Uh, Ul = split(U * S)
    xp = torch.zeros((N, F, 10, 10), dtype=torch.float32)
    xp[:, :, 1:9, 1:9] = x
    # input transform B^T d B in f32, rows then columns (two rounded additions per element)
    dr = [xp[:, :, k:k + 7:2, :] for k in range(4)]                   # (N,F,4,10): rows 2ty-1+k
    R = [dr[0] - dr[2], dr[1] + dr[2], dr[2] - dr[1], dr[1] - dr[3]]
    V = []
    for a in range(4):
        dc = [R[a][:, :, :, k:k + 7:2] for k in range(4)]             # (N,F,4,4)
        V.append([dc[0] - dc[2], dc[1] + dc[2], dc[2] - dc[1], dc[1] - dc[3]])
    M = [[None] * 4 for _ in range(4)]
    for a in range(4):
        for bb in range(4):
            vh, vl = split(V[a][bb].double())
            acc = torch.zeros((F, N * 16), dtype=torch.float32)
            for c in range(F // 16):
                bh = vh[:, c * 16:c * 16 + 16].permute(1, 0, 2, 3).reshape(16, N * 16)
                bl = vl[:, c * 16:c * 16 + 16].permute(1, 0, 2, 3).reshape(16, N * 16)
                acc = mfma3(acc, Uh[a][bb][:, c * 16:c * 16 + 16], Ul[a][bb][:, c * 16:c * 16 + 16], bh, bl)
            M[a][bb] = acc
    # output transform A^T M A in f32
    T = [[(M[0][bb] + M[1][bb]) + M[2][bb] for bb in range(4)], [(M[1][bb] - M[2][bb]) - M[3][bb] for bb in range(4)]]
    Y = [[(T[i][0] + T[i][1]) + T[i][2], (T[i][1] - T[i][2]) - T[i][3]] for i in range(2)]
    y = torch.stack([torch.stack(Y[0], dim=-1), torch.stack(Y[1], dim=-1)], dim=-2)      # (F, N*16, 2(i), 2(j))
    y = y.reshape(F, N, 4, 4, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(F, N, 8, 8) * np.float32(1.0 / S)
    return finish(y.permute(1, 0, 2, 3), b, skip)


def trunk(net, planes, conv):
    with torch.no_grad():
        x = torch.relu(net.stem(planes))
        hi, lo = split(x.double())
        x = (hi + lo).float()
        for c1, c2 in net.res:
            w1, b1 = net._fold(c1)
            w2, b2 = net._fold(c2)
            t = conv(x, w1, b1, None)
            x = conv(t, w2, b2, x)
    return x


def heads(net, x):
    with torch.no_grad():
        p = torch.relu(net.policy_conv(x)).flatten(1)
        p = torch.softmax(net.policy_fc(p), dim=1)
        v = torch.relu(net.value_conv(x)).flatten(1)
        v = torch.tanh(net.value_fc2(torch.relu(net.value_fc1(v))))
    return p, v[:, 0]


def positions(n, seed):
    """Random playouts frozen at a uniformly random ply (host scalar primitives of libraz)."""
    from reversi_alpha_zero_amd.lib import bitboard as bb
    rng = np.random.default_rng(seed)
    own_l, en_l = [], []
    while len(own_l) < n:
        black, white, player = 0x0000000810000000, 0x0000001008000000, 1
        stop = int(rng.integers(0, 59))
        for ply in range(60):
            own, enemy = (black, white) if player == 1 else (white, black)
            legal = bb.find_correct_moves(own, enemy)
            if not legal:
                own, enemy = enemy, own
                player = 3 - player
                legal = bb.find_correct_moves(own, enemy)
                if not legal:
                    break
            if ply == stop:
                own_l.append(own)
                en_l.append(enemy)
                break
            sq = [i for i in range(64) if legal >> i & 1]
            a = sq[int(rng.integers(0, len(sq)))]
            f = bb.calc_flip(a, own, enemy)
            own, enemy = own | f | (1 << a), enemy & ~f
            black, white = (own, enemy) if player == 1 else (enemy, own)
            player = 3 - player
    sh = np.arange(64, dtype=np.uint64)
    o = ((np.array(own_l, dtype=np.uint64)[:, None] >> sh) & 1).astype(np.float32)
    e = ((np.array(en_l, dtype=np.uint64)[:, None] >> sh) & 1).astype(np.float32)
    return torch.from_numpy(np.stack([o, e], axis=1).reshape(n, 2, 8, 8))


def stats(d):
    d = d.flatten().double()
    return {"max": float(d.max()), "mean": float(d.mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--variants", default="0,1,2")
    ap.add_argument("--models", default="direct,wino1d,wino2d")
    ap.add_argument("--net", default="256,10,256")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    F, R, V = (int(v) for v in a.net.split(","))
    planes = positions(a.n, 99)
    convs = {"direct": conv_direct, "wino1d": conv_wino1d, "wino2d": conv_wino2d}
    out = {"net": [F, R, V], "positions": a.n, "matrix_instruction_model": "16 exact products, one f32 rounding per instruction", "variants": []}
    for vi in (int(v) for v in a.variants.split(",")):
        name, seed, bn_seed, decades = VARIANTS[vi]
        net = ReversiNet(F, R, V).keras_init_(seed)
        if bn_seed is not None:
            net.randomize_bn_(bn_seed, decades=decades)
        net.eval()
        n64 = ReversiNet(F, R, V)
        n64.load_state_dict(net.state_dict())
        n64 = n64.double().eval()
        with torch.no_grad():
            dp, dv = n64(planes.double())
            tp, tv = net(planes)
        dv, tv = dv[:, 0], tv[:, 0]
        row = {"variant": name, "torch_fp32_vs_f64": {"policy": stats(tp.double() - dp), "value": stats(tv.double() - dv)}}
        row["torch_fp32_vs_f64"]["policy"]["max"] = float((tp.double() - dp).abs().max())
        row["torch_fp32_vs_f64"]["value"]["max"] = float((tv.double() - dv).abs().max())
        for m in a.models.split(","):
            p, v = heads(net, trunk(net, planes, convs[m]))
            row[m] = {"vs_f64": {"policy_max": float((p.double() - dp).abs().max()), "policy_mean": float((p.double() - dp).abs().mean()),
                                 "value_max": float((v.double() - dv).abs().max()), "value_mean": float((v.double() - dv).abs().mean())},
                      "vs_torch_fp32": {"policy_max": float((p - tp).abs().max()), "value_max": float((v - tv).abs().max())}}
            print(json.dumps({"variant": name, "model": m, **row[m]}), file=sys.stderr, flush=True)
        row["torch_fp32_vs_f64"]["policy"]["mean"] = float((tp.double() - dp).abs().mean())
        row["torch_fp32_vs_f64"]["value"]["mean"] = float((tv.double() - dv).abs().mean())
        out["variants"].append(row)
    text = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
