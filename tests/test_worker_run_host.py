"""BatchedSelfPlayWorker.run() (worker/self_play.py:95-137 of the reference) on the HOST: the control flow around the
engine - gather, resignation bookkeeping, broadcast, background file writing, game-index file - with the engine replaced
by a stub that hands out prepared records.  The real engine under run() is covered on the GPU (tests/test_multirank_gpu.py,
tests/test_worker_scale_gpu.py); this is the CPU (gloo, world_size 2) coverage of the N > 1 path."""
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port(_hint=None):
    """A port nobody listens on right now (the suite runs on several worker processes: fixed numbers collide)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def stub_games(first_id, n, seed=7):
    """Deterministic fake finished games keyed by GLOBAL game id (so that any sharding yields the same games)."""
    games = []
    for gid in range(first_id, first_id + n):
        rng = np.random.default_rng(seed * 100003 + gid)
        plies = []
        for j in range(int(rng.integers(3, 10))):
            own = int(rng.integers(0, 2**62)) * 2 + 1
            rn = [float(v) for v in rng.integers(0, 40, 64) * (rng.random(64) < 0.3)]
            if sum(rn) == 0:
                rn[5] = 3.0
            plies.append({"player": 1 + j % 2, "turn": j, "own": own, "enemy": int(rng.integers(0, 2**62)) & ~own, "action": int(rng.integers(0, 64)),
                          "has_row": bool(j % 4), "solved": False, "sims": 10, "loops": 1, "n": float(rng.integers(1, 9)),
                          "q": float(rng.random() * 2 - 1), "root_n": rn, "root_w": None})
        games.append((plies, {"winner": int(rng.integers(1, 4)), "status": 1, "plies": len(plies), "game_id": gid,
                              "enable_resign": int(gid % 2), "resigned_black": int(gid % 2), "resigned_white": int(gid % 5 == 0)}))
    return games


def make_stub_worker(cfg, games_in_flight, rank=0, world=1, seed=4):
    import torch
    from reversi_alpha_zero_amd.engine import GAME_SUMMARY, PLY_HEADER
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker

    class StubEngine:
        def __init__(self, games):
            self.games = games

        def pack_records(self, g0, n, plies=None):   # SelfPlayEngine.pack_records' result, as CPU tensors
            games = self.games[g0:g0 + n]
            ext = max(len(p) for p, _ in games) if plies is None else plies
            hdr = np.zeros((n, ext), dtype=PLY_HEADER)
            rn = np.zeros((n, ext, 64), dtype=np.uint32)
            sm = np.zeros(n, dtype=GAME_SUMMARY)
            for g, (pl, s) in enumerate(games):
                for j, p in enumerate(pl):
                    h = hdr[g, j]
                    h["own"], h["enemy"], h["n"], h["q"] = p["own"], p["enemy"], p["n"], p["q"]
                    h["action"], h["player"], h["turn"], h["has_row"] = p["action"], p["player"], p["turn"], int(p["has_row"])
                    rn[g, j] = np.asarray(p["root_n"], dtype=np.uint32)
                sm[g]["game_id"], sm[g]["n_plies"], sm[g]["status"] = s["game_id"], len(pl), s["winner"]
                sm[g]["resigned_black"], sm[g]["resigned_white"], sm[g]["enable_resign"] = s["resigned_black"], s["resigned_white"], s["enable_resign"]
            return {"headers": torch.from_numpy(hdr.view(np.uint8).reshape(n, ext, 48)), "root_n": torch.from_numpy(rn.view(np.int32)),
                    "summary": torch.from_numpy(sm.view(np.uint8).reshape(n, 32))}

    class StubNet:
        def range_ok(self):
            return True

    class StubWorker(BatchedSelfPlayWorker):
        thresholds_seen = []

        def play_batch_raw(self, first_game_idx=0, device_records=False):
            # a rank plays ids [first + rank * B, first + (rank + 1) * B)  (worker/self_play.py of this package: play_batch_raw)
            self._net = StubNet()
            self.thresholds_seen = self.thresholds_seen + [self.config.play.resign_threshold]
            first = first_game_idx + self.rank * self.games_in_flight
            return StubEngine(stub_games(first, self.games_in_flight)), self.games_in_flight

    return StubWorker(cfg, b"", games_in_flight=games_in_flight, seed=seed, device="cpu", rank=rank, world=world)


def say(*words):
    """One line of a rank's report as ONE write (two ranks share the launcher's stdout; print() writes its pieces separately)."""
    sys.stdout.write(" ".join(str(w) for w in words) + "\n")
    sys.stdout.flush()


def make_config(root):
    from reversi_alpha_zero_amd.config import Config
    cfg = Config()
    cfg.play_data.update(dict(nb_game_in_file=3, nb_game_in_ggf_file=4, drop_draw_game_rate=0.5, max_file_num=1000))
    cfg.play.resign_threshold = -0.8
    rc = cfg.resource
    rc.data_dir, rc.play_data_dir, rc.self_play_ggf_data_dir = str(root), str(root / "play"), str(root / "ggf")
    rc.self_play_game_idx_file = str(root / ".self-play-game-idx")
    rc.create_directories = lambda: [os.makedirs(d, exist_ok=True) for d in (rc.play_data_dir, rc.self_play_ggf_data_dir)]
    return cfg


def outputs(cfg):
    rc = cfg.resource
    play = [open(os.path.join(rc.play_data_dir, f), "rb").read() for f in sorted(os.listdir(rc.play_data_dir))]
    ggf = [re.sub(r"DT\[[^\]]*\]", "DT[]", open(os.path.join(rc.self_play_ggf_data_dir, f)).read())
           for f in sorted(os.listdir(rc.self_play_ggf_data_dir))]
    return play, ggf, open(rc.self_play_game_idx_file).read()


def test_run_background_writer_equals_inline(tmp_path):
    """run() with the files written by the background thread == written inline: same file contents in the same order,
    same game index, same threshold trajectory; rows load as JSON; memory-bounded streaming (ahead < games)."""
    res = {}
    for mode in (True, False):
        cfg = make_config(tmp_path / f"bg{int(mode)}")
        w = make_stub_worker(cfg, games_in_flight=50)
        w.run(total_games=250, background_emit=mode)
        res[mode] = outputs(cfg) + (w.thresholds_seen, cfg.play.resign_threshold, w.resign_test_game_count)
    assert res[True] == res[False]
    play, ggf, idx, thr, final_thr, _ = res[True]
    assert idx == "250" and len(play) >= 60 and len(ggf) >= 60
    assert len(thr) == 5 and len(set(thr)) > 1           # 25 no-resign test games per batch: the threshold moved after 100 of them
    rows = json.loads(play[0])
    assert len(rows) % 8 == 0 and len(rows[0]) == 3 and len(rows[0][1]) == 64
    # the streamed writer with a small look-ahead window writes the same bytes
    cfg = make_config(tmp_path / "ahead")
    w = make_stub_worker(cfg, games_in_flight=50)
    from reversi_alpha_zero_amd.engine import raw_from_packed
    eng, n = w.play_batch_raw(0)
    pk = eng.pack_records(0, n)
    raw = raw_from_packed(*(pk[k].numpy() for k in ("headers", "root_n", "summary")))
    cfg.resource.create_directories()
    w.write_raw(raw, 1, threads=2, ahead=3)
    first = [open(os.path.join(cfg.resource.play_data_dir, f), "rb").read() for f in sorted(os.listdir(cfg.resource.play_data_dir))]
    assert first == play[:len(first)] and len(first) >= 12


def test_run_reports_writer_errors(tmp_path):
    """A failure in the background writer surfaces in run() (not silently lost), and the game index is not advanced past
    the batch whose files are missing - whatever the interleaving of the caller and the writer thread."""
    cfg = make_config(tmp_path / "err")
    w = make_stub_worker(cfg, games_in_flight=20)
    real = w.write_raw
    calls = []

    def failing(raw, first_local_idx=1, threads=None, ahead=128):
        calls.append(first_local_idx)
        if len(calls) == 2:
            raise OSError("disk full")
        return real(raw, first_local_idx, threads, ahead)
    w.write_raw = failing
    with pytest.raises(RuntimeError, match="disk full"):
        w.run(total_games=100)
    assert open(cfg.resource.self_play_game_idx_file).read() == "20"
    assert calls == [1, 21]   # nothing queued behind the failed batch was written


def test_background_writer_drops_everything_behind_a_failed_batch():
    """The race the sticky flag closes, forced: the caller takes the exception out of the writer (its _check() at the next
    submit) BEFORE the thread looks at the batch queued behind the failed one; that batch must still be dropped."""
    import threading
    from reversi_alpha_zero_amd.worker.self_play import _BackgroundWriter
    written, idx = [], []
    failed_once, go_on = threading.Event(), threading.Event()

    class Stub:
        def write_raw(self, raw, local_idx):
            if raw == "bad":
                failed_once.set()
                raise OSError("disk full")
            go_on.wait(10)       # (only reached if the writer wrongly keeps going)
            written.append(raw)

        def _write_game_idx(self, game_idx):
            idx.append(game_idx)
    bw = _BackgroundWriter(Stub())
    bw.submit("bad", 1, 20)
    assert failed_once.wait(10)
    while not bw.failed:         # the thread has recorded the failure
        pass
    with pytest.raises(RuntimeError, match="disk full"):
        bw.submit("next", 21, 40)          # the caller is told here: this clears writer.error
    bw.queue.put(("late", 41, 60))         # a batch that slips in behind the failure (what the old flag let through)
    go_on.set()
    bw.close()                             # the error was already raised: close() has nothing new to report
    assert written == [] and idx == []
    with pytest.raises(RuntimeError, match="stopped"):
        bw.submit("again", 61, 80)


def test_block_is_replayed_on_f32_when_the_f16_range_flag_is_raised(tmp_path):
    """raz_net_range_check says an activation left the f16 range during a block: run() discards the block, rebuilds net and
    engine on the exact-f32 kernels (include/raz.h: "run the net with reserved = 0 then"), replays the SAME ids and goes on;
    the files are those of a run that never overflowed."""
    cfg = make_config(tmp_path / "ovf")
    w = make_stub_worker(cfg, games_in_flight=30)
    played, dropped = [], []

    class Net:
        def __init__(self, ok):
            self.ok = ok

        def range_ok(self):
            return self.ok
    stub_play = type(w).play_batch_raw

    def play(first_game_idx=0, device_records=False):
        out = stub_play(w, first_game_idx, device_records)
        played.append((first_game_idx, w._f32_fallback))
        w._net = Net(ok=not (first_game_idx == 30 and not w._f32_fallback))   # the second block overflows on the f16 path
        return out
    w.play_batch_raw = play
    w._drop_engine = lambda net_too=False: dropped.append(net_too)
    w.run(total_games=90)
    assert played == [(0, False), (30, False), (30, True), (60, True)] and dropped == [True]
    ref_cfg = make_config(tmp_path / "plain")
    make_stub_worker(ref_cfg, games_in_flight=30).run(total_games=90)
    assert outputs(cfg) == outputs(ref_cfg)
    # a second overflow - now on the f32 kernels - is an error, not a loop
    w2 = make_stub_worker(make_config(tmp_path / "ovf2"), games_in_flight=30)
    stub2 = type(w2).play_batch_raw

    def always_bad(first_game_idx=0, device_records=False):
        out = stub2(w2, first_game_idx, device_records)
        w2._net = Net(ok=False)
        return out
    w2.play_batch_raw = always_bad
    w2._drop_engine = lambda net_too=False: None
    with pytest.raises(RuntimeError, match="numeric range"):
        w2.run(total_games=30)


def test_rank0_decides_model_reloads(tmp_path):
    """reload_model is polled on rank 0 only and what it returns is what every rank loads (world 1: plain call)."""
    cfg = make_config(tmp_path / "reload")
    w = make_stub_worker(cfg, games_in_flight=10)
    seen, polls = [], []
    w.set_net_blob = lambda blob: seen.append(blob)

    def reload_model():
        polls.append(1)
        return b"gen2" if len(polls) == 2 else None
    w.run(total_games=30, reload_model=reload_model)
    assert len(polls) == 3 and seen == [b"gen2"]


def test_max_file_num_is_enforced_from_a_tracked_listing(tmp_path):
    """remove_play_data (self_play.py:209-217) with the directory listed once per batch: only the newest max_file_num
    files survive, older files of earlier runs included."""
    cfg = make_config(tmp_path / "cap")
    cfg.play_data.max_file_num = 7
    cfg.resource.create_directories()
    for i in range(3):
        open(os.path.join(cfg.resource.play_data_dir, cfg.resource.play_data_filename_tmpl % f"20000101-00000{i}.000000"), "w").write("[]")
    w = make_stub_worker(cfg, games_in_flight=40)
    w.run(total_games=40)
    names = sorted(os.listdir(cfg.resource.play_data_dir))
    assert len(names) == 7 and not any(n.startswith("play_2000") for n in names)


_RUN2_SCRIPT = r'''
import os, sys, pathlib
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch.distributed as dist
from test_worker_run_host import make_config, make_stub_worker, say
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = make_config(pathlib.Path({out!r}))
w = make_stub_worker(cfg, games_in_flight=25, rank=rank, world=world)
loaded, polls = [], []
w.set_net_blob = lambda blob: loaded.append(bytes(blob))
def reload_model():   # only rank 0 may be asked; what it answers is what BOTH ranks load
    assert rank == 0
    polls.append(1)
    return bytes(range(256)) * 40 if len(polls) == 2 else None
w.run(total_games=200, reload_model=reload_model)
assert loaded == [bytes(range(256)) * 40], (rank, len(loaded))
say("RANK", rank, "THRESHOLDS", w.thresholds_seen, cfg.play.resign_threshold)
dist.barrier(); dist.destroy_process_group()
'''


def test_run_two_ranks_gloo_equals_one_rank(tmp_path):
    """SURVEY 8(d) Config 4's acceptance on the host logic: 2 ranks x 25 games per batch write the files 1 rank x 50
    games per batch writes, and both ranks play every batch under the same (broadcast) resign threshold."""
    script = tmp_path / "run2.py"
    script.write_text(_RUN2_SCRIPT.format(root=ROOT, out=str(tmp_path / "two")))
    port = str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    thr = dict(re.findall(r"RANK (\d) THRESHOLDS (\[[^\]]*\] \S+)", r.stdout))
    assert set(thr) == {"0", "1"} and thr["0"] == thr["1"], r.stdout[-1000:]
    cfg1 = make_config(tmp_path / "one")
    w = make_stub_worker(cfg1, games_in_flight=50)
    w.run(total_games=200)
    two = outputs(make_config(tmp_path / "two"))
    one = outputs(cfg1)
    assert one[2] == two[2] == "200" and len(one[0]) >= 50
    assert [hashlib.sha256(b).hexdigest() for b in one[0]] == [hashlib.sha256(b).hexdigest() for b in two[0]] and one[1] == two[1]
    assert f"{w.thresholds_seen} {cfg1.play.resign_threshold}" == thr["0"]


_FATAL_SCRIPT = r'''
import os, sys, pathlib
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch.distributed as dist
from test_worker_run_host import make_config, make_stub_worker, say
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = make_config(pathlib.Path({out!r}))
w = make_stub_worker(cfg, games_in_flight=10, rank=rank, world=world)
stub_play = type(w).play_batch_raw
def play(first_game_idx=0, device_records=False):
    if rank == 1 and first_game_idx >= 20:      # the second block: rank 1's node pools are "full"
        raise RuntimeError("engine error flags 0x1 (1 node pool full)")
    return stub_play(w, first_game_idx, device_records)
w.play_batch_raw = play
try:
    w.run(total_games=60)
    say("RANK", rank, "RETURNED")
except RuntimeError as ex:
    say("RANK", rank, "RAISED", str(ex)[:60])
dist.barrier(); dist.destroy_process_group()
'''


def test_a_fatal_error_on_one_rank_raises_on_every_rank(tmp_path):
    """A genuine engine error (not the net's range flag) on ONE rank in the middle of a run: every rank learns of it in the
    block-state all_reduce and raises - the failing rank its own exception, the other one a RuntimeError naming the event - instead
    of the healthy rank waiting forever in the record gather (2 ranks over gloo; the whole run must end within the timeout)."""
    script = tmp_path / "fatal.py"
    script.write_text(_FATAL_SCRIPT.format(root=ROOT, out=str(tmp_path / "f")))
    port = str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RANK 1 RAISED engine error flags 0x1" in r.stdout and "RANK 0 RAISED rank 0: another rank failed" in r.stdout, r.stdout[-1500:]
    # and at world 1 the exception simply propagates
    w = make_stub_worker(make_config(tmp_path / "one"), games_in_flight=10)

    def boom(first_game_idx=0, device_records=False):
        raise RuntimeError("engine error flags 0x4 (4 records full)")
    w.play_batch_raw = boom
    with pytest.raises(RuntimeError, match="records full"):
        w.run(total_games=10)


# ---- per-rank emission: every rank writes the files of its own id range -------------------------------------------------------
_PER_RANK_SCRIPT = r'''
import os, sys, pathlib
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch.distributed as dist
from test_worker_run_host import make_config, make_stub_worker, say
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = make_config(pathlib.Path({out!r}))
cfg.play_data.update(dict(nb_game_in_file=5, nb_game_in_ggf_file=5, max_file_num={max_files}))
w = make_stub_worker(cfg, games_in_flight={in_flight}, rank=rank, world=world)
w.emission = {emission!r}
fail_at = {fail_at}
if fail_at and rank == {fail_rank}:
    real, calls = w.write_raw, []
    def failing(raw, first_local_idx=1, threads=None, ahead=128, stamp_base=None):
        calls.append(first_local_idx)
        if len(calls) == fail_at:
            import time; time.sleep(1.5)    # (rank 0's writer has the first block on disk by now: what the index may advance to)
            raise OSError("disk full")
        return real(raw, first_local_idx, threads, ahead, stamp_base)
    w.write_raw = failing
try:
    w.run(total_games={total}, background_emit={background})
    say("RANK", rank, "RETURNED")
except RuntimeError as ex:
    say("RANK", rank, "RAISED", str(ex)[:70].replace("\n", " "))
say("RANK", rank, "OWNFILES", int(w._per_rank_emission()), "BYTES", getattr(w, "bytes_written", 0), "GATHER", getattr(w, "last_gather_bytes", None))
say("RANK", rank, "THRESHOLDS", w.thresholds_seen, cfg.play.resign_threshold)
dist.barrier(); dist.destroy_process_group()
'''


def _per_rank_run(tmp_path, tag, port, emission="auto", background=True, max_files=1000, fail_at=0, nproc=2, in_flight=25, total=200, fail_rank=1):
    script = tmp_path / f"{tag}.py"
    port = _free_port(port)
    script.write_text(_PER_RANK_SCRIPT.format(root=ROOT, out=str(tmp_path / tag), emission=emission, background=background, max_files=max_files, fail_at=fail_at,
                                              in_flight=in_flight, total=total, fail_rank=fail_rank))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def _one_rank_reference(tmp_path, max_files=1000, in_flight=50, total=200):
    cfg = make_config(tmp_path / "one")
    cfg.play_data.update(dict(nb_game_in_file=5, nb_game_in_ggf_file=5, max_file_num=max_files))
    w = make_stub_worker(cfg, games_in_flight=in_flight)
    w.run(total_games=total)
    return outputs(cfg), f"{w.thresholds_seen} {cfg.play.resign_threshold}"


@pytest.mark.parametrize("background", [True, False])
def test_per_rank_emission_writes_the_files_of_the_rank0_path(tmp_path, background):
    """2 ranks x 25 ids per block with 5 games per file: "auto" lets every rank write the files of its own id range (only the
    32-byte summaries are gathered); the directory - play files and GGF files in name order, the game index - is what 1 rank x
    50 ids writes and what the same two ranks write with everything gathered on rank 0, byte for byte; both ranks play every
    block under the same broadcast threshold."""
    out = _per_rank_run(tmp_path, "two", 29561 + int(background), background=background)
    assert "RANK 0 RETURNED" in out and "RANK 1 RETURNED" in out, out[-1500:]
    own = dict(re.findall(r"RANK (\d) OWNFILES (\d) BYTES \d+ GATHER \d+", out))
    assert own == {"0": "1", "1": "1"}, out[-1500:]
    written = {k: int(v) for k, v in re.findall(r"RANK (\d) OWNFILES \d BYTES (\d+)", out)}
    assert written["0"] > 10000 and written["1"] > 10000                     # both ranks wrote files
    moved = {int(v) for v in re.findall(r"GATHER (\d+)", out)}
    assert moved == {25 * 32}                                                 # 32 B per game crossed, nothing else
    thr = dict(re.findall(r"RANK (\d) THRESHOLDS (\[[^\]]*\] \S+)", out))
    one, thr_one = _one_rank_reference(tmp_path)
    two = outputs(make_config(tmp_path / "two"))
    assert one[2] == two[2] == "200" and len(one[0]) >= 30 and len(one[1]) >= 40
    assert [hashlib.sha256(b).hexdigest() for b in one[0]] == [hashlib.sha256(b).hexdigest() for b in two[0]] and one[1] == two[1]
    assert thr["0"] == thr["1"] == thr_one
    if background:   # and the rank-0 path on the same two ranks
        out0 = _per_rank_run(tmp_path, "rank0", 29565, emission="rank0")
        assert dict(re.findall(r"RANK (\d) OWNFILES (\d)", out0)) == {"0": "0", "1": "0"}
        r0 = outputs(make_config(tmp_path / "rank0"))
        assert r0[0] == two[0] and r0[1] == two[1] and r0[2] == "200"


def test_per_rank_emission_keeps_the_newest_max_file_num_files(tmp_path):
    """max_file_num with two ranks pruning from their own listings: what is left at the end is exactly the newest files of the
    one-rank run (same contents, same order)."""
    out = _per_rank_run(tmp_path, "two", 29567, max_files=9)
    assert "RANK 0 RETURNED" in out and "RANK 1 RETURNED" in out, out[-1500:]
    one, _ = _one_rank_reference(tmp_path, max_files=9)
    two = outputs(make_config(tmp_path / "two"))
    assert len(two[0]) == 9 and two[0] == one[0]


def test_per_rank_emission_a_failed_writer_stops_every_rank(tmp_path):
    """Rank 1's writer fails on its second block: the failure travels in the next block-state all_reduce, BOTH ranks raise (no
    rank is left waiting in a collective), and data/.self-play-game-idx stays at the last block every rank has on disk (the first:
    rank 0 has written more by then, rank 1 never will)."""
    out = _per_rank_run(tmp_path, "fail", 29569, fail_at=2)
    assert "RANK 1 RAISED writing play data failed" in out and "disk full" in out, out[-1500:]
    assert "RANK 0 RAISED rank 0: another rank failed" in out, out[-1500:]
    idx = open(make_config(tmp_path / "fail").resource.self_play_game_idx_file).read()
    assert idx == "50", idx


def test_per_rank_emission_needs_whole_files_per_rank(tmp_path):
    """A rank's id range must be a whole number of files: "auto" falls back to the rank-0 path, "per_rank" says why it cannot."""
    cfg = make_config(tmp_path / "x")                     # 3 games per file, 4 per GGF file
    w = make_stub_worker(cfg, games_in_flight=25, rank=0, world=2)
    assert w._per_rank_emission() is False
    w.emission = "per_rank"
    with pytest.raises(ValueError, match="would span two ranks"):
        w._per_rank_emission()
    w = make_stub_worker(cfg, games_in_flight=24, rank=0, world=2)
    assert w._per_rank_emission() is True
    assert make_stub_worker(cfg, games_in_flight=24, rank=0, world=1)._per_rank_emission() is False
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    with pytest.raises(ValueError, match="emission"):
        BatchedSelfPlayWorker(cfg, b"", emission="everyone")


def test_whole_files_block_rounds_up_to_the_file_sizes(tmp_path):
    """start()'s default block under several ranks: a whole number of play files AND GGF files (ch5.yml: 1 and the default 100;
    mini.yml: 2 and 2), so that every rank writes its own files."""
    from reversi_alpha_zero_amd.worker.self_play import whole_files_block
    cfg = make_config(tmp_path / "x")
    cfg.play_data.update(dict(nb_game_in_file=1, nb_game_in_ggf_file=100, enable_ggf_data=True))
    assert whole_files_block(cfg, 16384) == 16400 and whole_files_block(cfg, 16400) == 16400
    cfg.play_data.enable_ggf_data = False
    assert whole_files_block(cfg, 16384) == 16384
    cfg.play_data.update(dict(nb_game_in_file=6, nb_game_in_ggf_file=4, enable_ggf_data=True))
    assert whole_files_block(cfg, 25) == 36
    w = make_stub_worker(cfg, games_in_flight=36, rank=1, world=2)
    assert w._per_rank_emission() is True


# ---- world 8: the rank count of the node the path is built for (BASELINE configs[3]: 65 536 games over 8 GPUs) ----------------------
def _sha(files):
    return [hashlib.sha256(b).hexdigest() for b in files]


def test_eight_ranks_rank0_emission_equals_one_rank(tmp_path):
    """8 ranks x 7 ids per block (7 is no whole number of 5-game files: "auto" gathers every record on rank 0, the path's one data
    collective at its full fan-in; 200 requested games are no whole number of 56-id blocks either: the run plays 4 blocks = 224 ids, as
    one rank x 56 does): the directory, the game index and the threshold trajectory are those of one rank."""
    out = _per_rank_run(tmp_path, "eight", 29571, nproc=8, in_flight=7)
    assert all(f"RANK {r} RETURNED" in out for r in range(8)), out[-2000:]
    assert dict(re.findall(r"RANK (\d) OWNFILES (\d)", out)) == {str(r): "0" for r in range(8)}
    written = {k: int(v) for k, v in re.findall(r"RANK (\d) OWNFILES \d BYTES (\d+)", out)}
    assert written["0"] > 10000 and all(written[str(r)] == 0 for r in range(1, 8))      # rank 0 alone wrote
    thr = dict(re.findall(r"RANK (\d) THRESHOLDS (\[[^\]]*\] \S+)", out))
    one, thr_one = _one_rank_reference(tmp_path, in_flight=56)
    eight = outputs(make_config(tmp_path / "eight"))
    assert one[2] == eight[2] == "224" and len(one[0]) >= 35
    assert _sha(one[0]) == _sha(eight[0]) and one[1] == eight[1]
    assert len(set(thr.values())) == 1 and thr["0"] == thr_one and len(thr) == 8


def test_eight_ranks_per_rank_emission_equals_one_rank(tmp_path):
    """8 ranks x 10 ids per block, 5 games per file: every rank writes the two files (and the GGF files) of its own id range, 32 B per
    game are gathered; 240 games = 3 blocks.  Byte-identical to one rank x 80, and max_file_num pruned from eight listings leaves the
    newest files of the one-rank run."""
    out = _per_rank_run(tmp_path, "eight", 29573, nproc=8, in_flight=10, total=240)
    assert all(f"RANK {r} RETURNED" in out for r in range(8)), out[-2000:]
    assert dict(re.findall(r"RANK (\d) OWNFILES (\d)", out)) == {str(r): "1" for r in range(8)}
    written = {k: int(v) for k, v in re.findall(r"RANK (\d) OWNFILES \d BYTES (\d+)", out)}
    assert all(written[str(r)] > 3000 for r in range(8)), written
    assert {int(v) for v in re.findall(r"GATHER (\d+)", out)} == {7 * 10 * 32}    # 32 B per game of the seven other ranks: nothing else crossed
    thr = dict(re.findall(r"RANK (\d) THRESHOLDS (\[[^\]]*\] \S+)", out))
    one, thr_one = _one_rank_reference(tmp_path, in_flight=80, total=240)
    eight = outputs(make_config(tmp_path / "eight"))
    assert one[2] == eight[2] == "240" and len(one[0]) >= 35
    assert _sha(one[0]) == _sha(eight[0]) and one[1] == eight[1]
    assert len(set(thr.values())) == 1 and thr["0"] == thr_one
    # pruning
    out = _per_rank_run(tmp_path, "pruned", 29575, nproc=8, in_flight=10, total=240, max_files=11)
    assert all(f"RANK {r} RETURNED" in out for r in range(8)), out[-2000:]
    cfgp = make_config(tmp_path / "one_pruned")
    cfgp.play_data.update(dict(nb_game_in_file=5, nb_game_in_ggf_file=5, max_file_num=11))
    make_stub_worker(cfgp, games_in_flight=80).run(total_games=240)
    pruned = outputs(make_config(tmp_path / "pruned"))
    assert len(pruned[0]) == 11 and _sha(pruned[0]) == _sha(outputs(cfgp)[0])


def test_eight_ranks_a_failed_writer_on_rank_5_stops_every_rank(tmp_path):
    """Rank 5's writer fails on its second block: all eight ranks raise within the run (nobody waits in a collective), rank 5 its own
    exception, and the game index stays at the one block every rank has on disk."""
    out = _per_rank_run(tmp_path, "fail", 29577, nproc=8, in_flight=10, total=240, fail_at=2, fail_rank=5)
    assert "RANK 5 RAISED writing play data failed" in out and "disk full" in out, out[-2000:]
    for r in (0, 1, 2, 3, 4, 6, 7):
        assert f"RANK {r} RAISED rank {r}: another rank failed" in out, out[-2000:]
    assert open(make_config(tmp_path / "fail").resource.self_play_game_idx_file).read() == "80"


def test_auto_emission_across_hosts_gathers_on_rank0(tmp_path, monkeypatch):
    """emission="auto" picks per-rank files only when all ranks run on ONE host (the launcher's LOCAL_WORLD_SIZE == the world size):
    on several nodes each rank would write into its own node's directory.  "per_rank" stays the caller's explicit choice."""
    cfg = make_config(tmp_path / "x")
    cfg.play_data.update(dict(nb_game_in_file=5, nb_game_in_ggf_file=5))
    w = make_stub_worker(cfg, games_in_flight=10, rank=3, world=16)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert w._per_rank_emission() is False
    w.emission = "per_rank"
    assert w._per_rank_emission() is True
    w.emission = "auto"
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "16")
    assert w._per_rank_emission() is True
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert w._per_rank_emission() is True


def test_start_block_of_ch5_yml_on_eight_ranks_is_whole_files(tmp_path):
    """start()'s default block under 8 ranks with ch5.yml's file sizes (1 game per play file, 100 per GGF file): 4 x 4096 = 16 384 ids
    per rank is rounded up to 16 400, so that every one of the 8 ranks owns whole files and writes them itself."""
    from reversi_alpha_zero_amd.worker.self_play import whole_files_block
    cfg = make_config(tmp_path / "x")
    cfg.play_data.update(dict(nb_game_in_file=1, nb_game_in_ggf_file=100, enable_ggf_data=True))
    blk = whole_files_block(cfg, 4 * 4096)
    assert blk == 16400
    for rank in range(8):
        w = make_stub_worker(cfg, games_in_flight=4096, rank=rank, world=8)
        w.block_games = blk
        assert w._ids_per_block() == blk and w._per_rank_emission() is True
