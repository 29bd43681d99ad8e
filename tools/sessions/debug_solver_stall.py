"""GPU diagnostics: mini.yml as shipped on the configs[1] batch (two-kernel pipeline) on a workspace that starts as GARBAGE (a run after
other engines of the process); when the batch stops making progress, dump what the solver pool holds."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402

g.build()
from reversi_alpha_zero_amd._native import lib  # noqa: E402
from reversi_alpha_zero_amd.agent.model import ReversiNet  # noqa: E402
from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine  # noqa: E402

dev = torch.device("cuda:0")
if "--dirty" in sys.argv:
    junk = [torch.full((1 << 30,), 0xA5, dtype=torch.uint8, device=dev) for _ in range(40)]
    del junk
games, sims = 4096, 200
cfg = bench.mini_config(sims, 4)
cfg.play.thinking_loop, cfg.play.use_solver_turn, cfg.play.use_solver_turn_in_simulation = 2, 50, 50
blob = ReversiNet(*bench.NETS["mini"]).keras_init_(0).to_blob()
eng = SelfPlayEngine(cfg, DeviceNet(blob, dev), n_games=games, seed=0, sims_hint=2 * sims, fused=False)
eng.start(0, sims)
eng.step(50)
eng.stats()
eng.start(0, sims)
WS = 64 + 12288 + 135168
last, still, steps = -1, 0, 0
while steps < 40000:
    eng.step(200)
    steps += 200
    st = eng.stats()
    if eng.pool_nearly_full(st, 200):
        eng.gc(min(eng.cfg.nodes_per_game // 4, st["max_pool_used"] // 2))
    if st["finished_games"] >= games:
        print(json.dumps({"finished": True, "steps": steps}))
        sys.exit(0)
    still = still + 1 if st["total_sims"] == last else 0
    last = st["total_sims"]
    if still >= 5:
        break
print(json.dumps({"stalled_at_step": steps, "finished_games": st["finished_games"], "total_sims": st["total_sims"]}))


def read(which, off, n):
    buf = (ctypes.c_ubyte * n)()
    assert lib.raz_engine_debug_read(eng._h, which, off, n, buf) == 0
    return np.frombuffer(buf, dtype=np.uint8).copy()


hdrs = np.stack([read(6, i * WS, 64).view(np.uint32) for i in range(games)])
state = hdrs[:, 0]
print("states", {int(k): int((state == k).sum()) for k in np.unique(state)})
run = np.nonzero((state == 2) | (state == 1))[0]
ks = ["state", "gen", "own_lo", "own_hi", "en_lo", "en_hi", "exact", "k_n2", "tasks", "total", "next", "ans_move", "limit", "posted", "rounds", "rounds_total"]
for i in run[:6]:
    h = hdrs[i]
    d = dict(zip(ks, [int(x) for x in h]))
    tree = read(6, int(i) * WS + 64, 12288)
    deep = read(6, int(i) * WS + 64 + 12288, 135168)
    T = d["tasks"]
    res_off = 3 * 8 * 14 + 3 * 8 * 182 + 2 * 184 + 16
    result = tree[res_off:res_off + T].view(np.int8)
    sub_first = deep[3 * 8 * 2184:3 * 8 * 2184 + 2 * (T + 1)].view(np.uint16)
    kind_off = 3 * 8 * 2184 + 2 * 2188 + 2184   # (h_dead lies before h_kind)
    h_kind = deep[kind_off:kind_off + T]
    sr_off = kind_off + 2184
    sub_result = deep[sr_off:sr_off + d["total"]].view(np.int8)
    unk = np.nonzero(sub_result == -128)[0]
    print("game", int(i), d, "unknown level-3 results", int((result == -128).sum()), "of", T, "unknown task results", len(unk), "first", unk[:10].tolist(),
          "kinds", {int(k): int((h_kind == k).sum()) for k in np.unique(h_kind)}, "sub_first tail", sub_first[-3:].tolist())
W = int(eng.cfg.solver_pool_waves) or min(1280, (games + 3) // 4)
have, held = 0, {}
for w in range(W):
    words = read(7, w * 8192, 8192).view(np.uint64).reshape(16, 64)
    m1 = words[3]
    hv = (m1 & 1) != 0
    have += int(hv.sum())
    for ln in np.nonzero(hv)[0]:
        gslot = int(words[4][ln] & 0xffffffff)
        held.setdefault(gslot, []).append((w, int(ln), int(words[5][ln] & 0xffff), int((m1[ln] >> 8) & 0xff)))
print("worker lanes with a search in hand:", have, "games they belong to:", len(held))
for i in run[:6]:
    print("game", int(i), "held by", held.get(int(i), [])[:8])
print("pool headers", read(8, 0, 8 * 64).view(np.uint32).reshape(8, 16)[:3, :2].tolist())
