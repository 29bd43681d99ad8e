"""GPU: the N > 1 path of the `self` worker on ONE GPU - two ranks share cuda:0 and rendezvous over gloo (a test rig:
the collectives' tensors are staged through host memory; under nccl the same code gathers straight from HBM) - and the
engine's device-side record packing.

SURVEY.md 8(d) "Config 4" acceptance check, scaled down: the files rank 0 writes for a 2-rank run (rank r plays ids
[r*B, (r+1)*B), one gather per batch) are byte-identical to the files of a 1-rank run over the same global game ids."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle as O
from conftest import ROOT
from oracle_util import load_mcts_golden, golden_net_blob, config_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

_WORKER_SCRIPT = '''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch, torch.distributed as dist
from oracle_util import load_mcts_golden, golden_net_blob
from reversi_alpha_zero_amd.config import Config
from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
elif os.environ.get("RAZ_TEST_NCCL_WORLD1") == "1":   # the nccl (RCCL) code path on the one GPU there is: a group of one rank
    os.environ.setdefault("MASTER_PORT", "29557")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
gold = load_mcts_golden()
g0 = next(g for g in gold["games"] if g["variant"] == "mini_shared")
cfg = Config()
cfg.play.update(g0["resolved_play"])
cfg.play.schedule_of_simulation_num_per_move = [(0, 10)]
cfg.play.resign_threshold, cfg.play.disable_resignation_rate = -0.3, 0.5     # resignations AND test games in a small batch
cfg.play.resign_threshold_delta, cfg.play.false_positive_threshold = 0.05, 0.0
cfg.play_data.update(dict(g0["resolved_play_data"], nb_game_in_file=5, nb_game_in_ggf_file=7, drop_draw_game_rate=0.5))
rc = cfg.resource
out = {out!r}
rc.data_dir = out; rc.play_data_dir = os.path.join(out, "play_data"); rc.self_play_ggf_data_dir = os.path.join(out, "ggf")
rc.model_dir = os.path.join(out, "model"); rc.next_generation_model_dir = os.path.join(out, "model", "next")
rc.log_dir = os.path.join(out, "logs"); rc.project_dir = out
rc.force_simulation_num_file = os.path.join(out, ".force-sim"); rc.self_play_game_idx_file = os.path.join(out, ".self-play-game-idx")
B = {per_rank}
w = BatchedSelfPlayWorker(cfg, golden_net_blob(gold["net"]), games_in_flight=B, seed=21, device="cuda:0", rank=rank, world=world)
# make every batch update the threshold (>= 100 test games in the reference; scaled down for the test)
orig = w.check_and_update_resignation_threshold
def check():
    if w.resign_test_game_count >= 4:
        w.resign_test_game_count += 100
        orig()
w.check_and_update_resignation_threshold = check
w.run(total_games={total})
print("THRESHOLD", rank, repr(cfg.play.resign_threshold))
print("GATHER", rank, getattr(w, "last_gather_backend", None), getattr(w, "last_gather_bytes", None))
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
'''


def _run(tmp_path, tag, world, per_rank, total, nccl_world1=False):
    out = tmp_path / tag
    out.mkdir()
    script = tmp_path / f"{tag}.py"
    script.write_text(_WORKER_SCRIPT.format(root=ROOT, out=str(out), per_rank=per_rank, total=total))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RAZ_TEST_NCCL_WORLD1"):
        env.pop(k, None)
    if nccl_world1:
        env["RAZ_TEST_NCCL_WORLD1"] = "1"
    if world == 1:
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29551", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    files = sorted(os.listdir(out / "play_data"))
    ggf = sorted(os.listdir(out / "ggf"))
    thr = sorted(line for line in r.stdout.splitlines() if line.startswith("THRESHOLD"))
    if nccl_world1:
        assert "GATHER 0 nccl" in r.stdout, r.stdout[-1500:]
    return ([open(out / "play_data" / f).read() for f in files], [open(out / "ggf" / f).read() for f in ggf],
            open(out / ".self-play-game-idx").read(), thr)


def test_two_rank_worker_files_equal_one_rank_files(tmp_path):
    """2 ranks x 6 slots x 2 batches vs 1 rank x 12 slots x 2 batches: play_*.json and GGF files byte-identical (the GGF
    header's date aside), game index equal, and both ranks end with the SAME resign threshold as the 1-rank run (the
    threshold is stepped on rank 0 and broadcast - every batch here - so the second batch's games depend on it)."""
    one = _run(tmp_path, "one", 1, 12, 24)
    two = _run(tmp_path, "two", 2, 6, 24)
    assert len(one[0]) >= 3 and one[0] == two[0]
    import re
    norm = lambda texts: [re.sub(r"DT\[[^\]]*\]", "DT[]", t) for t in texts]
    assert len(one[1]) >= 2 and norm(one[1]) == norm(two[1])
    assert one[2] == two[2] == "24"
    t1 = {l.split()[2] for l in one[3]}
    t2 = {l.split()[2] for l in two[3]}
    assert len(two[3]) == 2 and len(t2) == 1 and t1 == t2, (one[3], two[3])
    assert t1 != {repr(-0.3)}, "the threshold never moved: the test does not exercise the broadcast"


def test_worker_on_an_nccl_group_of_one_rank_writes_the_same_files(tmp_path):
    """The nccl (RCCL) branch of the worker on the one GPU a test box has: BatchedSelfPlayWorker.run inside
    init_process_group("nccl", world_size=1) takes every collective of the N > 1 path - the block-state all_reduce, gather_packed
    with device tensors straight from HBM (all_reduce of the ply extent, gather), broadcast_object_list of game index and resign
    threshold, barriers - and must write exactly the files of a run without a process group."""
    plain = _run(tmp_path, "plain", 1, 12, 24)
    nccl = _run(tmp_path, "nccl1", 1, 12, 24, nccl_world1=True)
    import re
    norm = lambda texts: [re.sub(r"DT\[[^\]]*\]", "DT[]", t) for t in texts]
    assert len(plain[0]) >= 3 and plain[0] == nccl[0] and norm(plain[1]) == norm(nccl[1]) and plain[2] == nccl[2] == "24"
    assert plain[3] == nccl[3]


def test_pack_records_equals_read_records():
    """raz_engine_pack_records (dense, ply-trimmed, device-to-device) == raz_engine_read_records field by field, for
    the whole batch and for a sub-range packed to a longer extent."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine, raw_from_packed
    gold = load_mcts_golden()
    g0 = next(g for g in gold["games"] if g["variant"] == "agz_resign")
    cfg = config_of(g0)
    eng = SelfPlayEngine(cfg, DeviceNet(golden_net_blob(gold["net"]), DEV), n_games=20, seed=4, sims_hint=10)
    eng.start(first_game_id=700, sims_per_move=10)
    eng.run(chunk=128)
    raw = eng.read_raw()
    mp = int(raw["n_plies"].max())
    for first, n, plies in ((0, 20, None), (5, 9, mp + 3)):
        pk = eng.pack_records(first, n, plies)
        got = raw_from_packed(*(pk[k].cpu().numpy() for k in ("headers", "root_n", "summary")))
        ext = plies or mp
        assert got["headers"].shape == (n, ext) and pk["headers"].device.type == "cuda"
        sl = slice(first, first + n)
        assert np.array_equal(got["headers"], raw["headers"][sl, :ext]) or all(
            np.array_equal(got["headers"][g, :int(raw["n_plies"][first + g])], raw["headers"][first + g, :int(raw["n_plies"][first + g])])
            for g in range(n))
        for g in range(n):
            k = int(raw["n_plies"][first + g])
            assert np.array_equal(got["root_n"][g, :k], raw["root_n"][first + g, :k])
            assert not got["root_n"][g, k:].any() and not got["headers"][g, k:].view(np.uint8).any()
        for key in ("n_plies", "status", "resigned", "game_id", "enable_resign", "final_black", "final_white"):
            assert np.array_equal(got[key], raw[key][sl]), key


def test_set_resign_threshold_at_run_time():
    """raz_engine_set_resign_threshold: the same engine, restarted on the same ids, plays the games of an engine CREATED
    with that threshold (and the slot's tree machinery is untouched: no rebuild)."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    gold = load_mcts_golden()
    g0 = next(g for g in gold["games"] if g["variant"] == "agz_resign")
    blob = golden_net_blob(gold["net"])
    cfg = config_of(g0)
    cfg.play.resign_threshold = -0.5
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=16, seed=6, sims_hint=12)
    eng.start(0, 12)
    eng.run(chunk=128)
    a = eng.records()
    eng.set_resign_threshold(-0.02)
    eng.start(0, 12)
    eng.run(chunk=128)
    b = eng.records()
    cfg.play.resign_threshold = -0.02
    ref = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=16, seed=6, sims_hint=12)
    ref.start(0, 12)
    ref.run(chunk=128)
    c = ref.records()
    assert b == c and a != b
    ocfg = O.play_cfg_from_config(cfg)
    for i in (0, 7, 15):
        plies, summ = O.selfplay_game(ocfg, blob, 6, i, 12)
        assert [p["action"] for p in plies] == [p["action"] for p in b[i][0]] and summ["winner"] == b[i][1]["winner"]
