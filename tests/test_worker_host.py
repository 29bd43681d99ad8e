"""CPU: host-side logic of the self-play worker — training-row emission (8 symmetries, z labels,
file order), GGF move strings, per-game simulation schedule, config overlay, and the multi-rank
record gather (world_size 2 over gloo).  No kernel is launched."""
import hashlib
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from conftest import ROOT, H


def _free_port():
    """A port nobody listens on right now, as a string (the suite runs on several worker processes: fixed numbers collide)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])
from oracle_util import load_mcts_golden, dense


def _plies_of(g):
    return [dict(p, own=int(p["own"], 16), enemy=int(p["enemy"], 16), root_n=dense(p["root_n"]),
                 saved_policy=dense(p["saved_policy"]) if p["has_row"] else None) for p in g["plies"]]


def test_rows_identical_to_reference_play_files():
    """rows_of_game (libraz host symmetries + numpy policy symmetries) reproduces the play_*.json
    text the reference worker wrote, byte for byte (sha256 of json.dumps)."""
    from reversi_alpha_zero_amd.worker.self_play import rows_of_game
    for g in load_mcts_golden()["games"]:
        if g["play_rows_sha256"] is None:
            continue
        rows = rows_of_game(_plies_of(g), g["winner"])
        assert len(rows) == g["play_rows_count"]
        assert hashlib.sha256(json.dumps(rows).encode()).hexdigest() == g["play_rows_sha256"], g["variant"]


def test_sym8_rows_vs_reference_test_case(golden_bb):
    """agent/player.py:166-179 incl. the reference's own test (test/agent/test_player.py:11-75)."""
    from reversi_alpha_zero_amd.worker.self_play import rows_of_game
    for case in golden_bb["sym8_rows"]:
        pol = np.zeros(64)
        for k, v in case["policy"].items():
            pol[int(k)] = v
        ply = {"own": H(case["own"]), "enemy": H(case["enemy"]), "player": 1, "has_row": True, "saved_policy": list(pol)}
        rows = rows_of_game([ply], winner=3)
        assert [[r[0][0], r[0][1], r[1]] for r in rows] == [[H(o), H(e), p] for o, e, p in case["rows"]]


def test_ggf_moves_vs_reference():
    from reversi_alpha_zero_amd.worker.self_play import ggf_moves_of_game
    from reversi_alpha_zero_amd.lib.ggf import convert_action_to_move, convert_move_to_action, make_ggf_string
    for g in load_mcts_golden()["games"]:
        assert ggf_moves_of_game(_plies_of(g)) == g["ggf_moves"], g["variant"]
    # test/lib/test_ggf.py:32-43
    assert convert_move_to_action("A1") == 0 and convert_move_to_action("H8") == 63
    assert convert_move_to_action("F5") == 44 and convert_move_to_action("PA") is None
    assert convert_action_to_move(0) == "A1" and convert_action_to_move(44) == "F5" and convert_action_to_move(None) == "PA"
    s = make_ggf_string("RAZ", "RAZ", moves=["C4/1.0/2.0", "PA"])
    assert s.startswith("(;GM[Othello]PC[RAZSelf]DT[") and s.endswith("B[C4/1.0/2.0]W[PA];)")


def test_schedule_and_force_sim(tmp_path):
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker.self_play import decide_simulation_num_per_move
    cfg = Config()
    cfg.resource.force_simulation_num_file = str(tmp_path / ".force-sim")
    assert [decide_simulation_num_per_move(cfg, i) for i in (0, 299, 300, 1999, 2000, 10**6)] == [8, 8, 50, 50, 200, 200]
    (tmp_path / ".force-sim").write_text("123\n")
    assert decide_simulation_num_per_move(cfg, 0) == 123
    # the worker's block form: one array per block, the force file read once
    from reversi_alpha_zero_amd.worker.self_play import simulation_nums_of_ids, default_block_games
    assert simulation_nums_of_ids(cfg, 290, 20).tolist() == [123] * 20
    (tmp_path / ".force-sim").unlink()
    ids = list(range(290, 310)) + [1999, 2000]
    assert simulation_nums_of_ids(cfg, 290, 20).tolist() == [decide_simulation_num_per_move(cfg, i) for i in range(290, 310)]
    assert simulation_nums_of_ids(cfg, 1999, 2).tolist() == [50, 200] and simulation_nums_of_ids(cfg, 0, 0).size == 0
    cfg.play.schedule_of_simulation_num_per_move = [(5, 9)]
    with pytest.raises(ValueError, match="no entry for game index 3"):
        simulation_nums_of_ids(cfg, 3, 4)
    # default block: 4 games per slot for wide nets, 16 for 16-filter nets (worker.start)
    import struct
    blob = lambda f: struct.pack("<8i", 0x4E5A4152, 1, f, 1, f, 3, 0, 0)
    assert default_block_games(blob(16), 4096) == 65536 and default_block_games(blob(256), 8192) == 32768 and default_block_games(None, 100) == 400


@pytest.mark.needs_reference
def test_config_overlay_matches_reference_yml():
    """Loading the reference's own yml files gives the same play/play_data/model values as the
    reference Config overlay."""
    import ref_harness as rh
    from reversi_alpha_zero_amd.config import load_config
    for yml in ("mini.yml", "ch5.yml", "alpha_go_zero.yml"):
        mine = load_config(os.path.join(rh.REFERENCE_ROOT, "config", yml))
        ref = rh.load_config(yml)
        for sec in ("play", "play_data", "model"):
            a, b = getattr(mine, sec), getattr(ref, sec)
            for k, v in vars(b).items():
                got = getattr(a, k)
                assert (list(map(list, got)) if k.startswith("schedule") else got) == \
                       (list(map(list, v)) if k.startswith("schedule") else v), (yml, sec, k)


def test_host_rng_matches_oracle():
    import oracle as O
    from reversi_alpha_zero_amd._rng import rng_pair
    for args in [(0, 0, 0, 0), (7, 3, 3, 0), (123, 99, 1, 5, 2, 3), (2**32 - 1, 2**32 - 1, 2, 10**6, 33, 7)]:
        assert rng_pair(*args) == O.rng_pair(*args)


def test_pack_unpack_roundtrip():
    from reversi_alpha_zero_amd.worker.self_play import pack_records, unpack_records
    recs = _fake_records(0, 5)
    back = unpack_records(pack_records(recs))
    assert [[{k: v for k, v in p.items() if k != "root_w"} for p in pl] for pl, _ in back] == \
           [[{k: v for k, v in p.items() if k != "root_w"} for p in pl] for pl, _ in recs]
    assert [s for _, s in back] == [s for _, s in recs]


def _fake_records(rank, n):
    rng = np.random.default_rng(100 + rank)
    out = []
    for i in range(n):
        npl = int(rng.integers(1, 9))
        plies = []
        for j in range(npl):
            rn = [float(v) for v in rng.integers(0, 50, 64)]
            plies.append({"player": 1 + j % 2, "turn": j, "own": int(rng.integers(0, 2**63)) * 2 + 1,
                          "enemy": int(rng.integers(0, 2**63)), "action": int(rng.integers(-1, 64)),
                          "has_row": bool(j % 3), "solved": bool(j % 5 == 4), "sims": 20, "loops": 1, "n": float(rng.integers(0, 9)),
                          "q": float(rng.random()), "root_n": rn, "root_w": None,
                          "saved_policy": [v / max(sum(rn), 1.0) for v in rn]})
        out.append((plies, {"winner": int(rng.integers(1, 4)), "status": 1, "plies": npl, "game_id": rank * 1000 + i,
                            "enable_resign": 1, "resigned_black": 0, "resigned_white": int(i % 2),
                            "black": int(rng.integers(0, 2**63)) * 2 + 1, "white": int(rng.integers(0, 2**63))}))
    return out


_GATHER_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from test_worker_host import _fake_records
from reversi_alpha_zero_amd.worker.self_play import gather_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = _fake_records(rank, 3 + 2 * rank)          # ragged: ranks hold different numbers of games/plies
got = gather_records(mine, rank, world)
if rank == 0:
    exp = _fake_records(0, 3) + _fake_records(1, 5)
    strip = lambda recs: [[[{{k: v for k, v in p.items() if k != "root_w"}} for p in pl], s] for pl, s in recs]
    assert strip(got) == strip(exp), "gathered records differ"
    print("GATHER_OK", len(got))
else:
    assert got == []
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_records_two_ranks_gloo(tmp_path):
    """The only collective of the path, at world_size 2 on CPU (gloo), ragged per-rank sizes."""
    script = tmp_path / "gather2.py"
    script.write_text(_GATHER_SCRIPT.format(root=ROOT))
    _p = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_p)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _p, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK 8" in r.stdout


def test_convert_to_training_data_equals_reference_recipe():
    """lib/data_helper.convert_to_training_data == the reference's per-row bit_to_array recipe
    (worker/optimize.py:214-231) on golden rows, including boards with bit 63 set."""
    import numpy as np
    from reversi_alpha_zero_amd.lib.data_helper import convert_to_training_data, pack_game_data
    from reversi_alpha_zero_amd.lib.bitboard import bit_to_array
    rows = [[[0x0000000810000000, 0x0000001008000000], [1 / 64] * 64, 1],
            [[0x8000000000000001, 0x7ffffffffffffffe], [0.0] * 63 + [1.0], -1],
            [[0, 0xffffffffffffffff], [0.5, 0.5] + [0.0] * 62, 0]]
    state, policy, z = convert_to_training_data(rows)
    exp_state = np.array([[bit_to_array(r[0][0], 64).reshape(8, 8), bit_to_array(r[0][1], 64).reshape(8, 8)] for r in rows])
    assert state.shape == (3, 2, 8, 8) and np.array_equal(state, exp_state)
    assert np.array_equal(policy, np.array([r[1] for r in rows])) and list(z) == [1, -1, 0]
    own, enemy, pol32, z8 = pack_game_data(rows)
    assert own.dtype == np.uint64 and int(own[1]) == 0x8000000000000001 and pol32.dtype == np.float32 and z8.dtype == np.int8


def test_evaluate_config_and_helpers(tmp_path):
    """EvaluateConfig mirrors config.py:101-110; next-generation dir listing and best-model save/load helpers."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.lib.data_helper import get_next_generation_model_dirs
    from reversi_alpha_zero_amd.lib.model_helpler import load_best_model_weight, save_as_best_model, reload_best_model_weight_if_changed
    from reversi_alpha_zero_amd.agent.model import ReversiModel
    cfg = Config()
    assert (cfg.eval.game_num, cfg.eval.replace_rate, cfg.eval.play_config.simulation_num_per_move,
            cfg.eval.play_config.noise_eps, cfg.eval.play_config.change_tau_turn) == (200, 0.55, 400, 0, 0)
    rc = cfg.resource
    rc.model_dir = str(tmp_path / "model")
    rc.model_best_config_path = str(tmp_path / "model" / "model_best_config.json")
    rc.model_best_weight_path = str(tmp_path / "model" / "model_best_weight.h5")
    rc.next_generation_model_dir = str(tmp_path / "model" / "next_generation")
    import os
    os.makedirs(rc.next_generation_model_dir)
    for name in ("model_20260101-000000.000000", "model_20260102-000000.000000"):
        os.makedirs(os.path.join(rc.next_generation_model_dir, name))
    assert [os.path.basename(d) for d in get_next_generation_model_dirs(rc)] == \
        ["model_20260101-000000.000000", "model_20260102-000000.000000"]
    cfg.model.update(dict(cnn_filter_num=16, res_layer_num=1, value_fc_size=16))
    m = ReversiModel(cfg)
    assert not load_best_model_weight(m)
    m.build(seed=3)
    save_as_best_model(m)
    m2 = ReversiModel(cfg)
    assert load_best_model_weight(m2) and m2.model.to_blob() == m.model.to_blob()
    assert reload_best_model_weight_if_changed(m2) is False


# ---- native row emission (csrc/raz_emit.hip) and the raw-array path of the worker -------------------------
def _raw_of_games(games, max_plies=72):
    """Engine-style record arrays (SelfPlayEngine.read_raw) of oracle / golden games: [(plies, summary)]."""
    from reversi_alpha_zero_amd.engine import PLY_HEADER
    n = len(games)
    raw = {"headers": np.zeros((n, max_plies), dtype=PLY_HEADER), "root_n": np.zeros((n, max_plies, 64), dtype=np.uint32),
           "n_plies": np.zeros(n, dtype=np.uint32), "status": np.zeros(n, dtype=np.uint8),
           "resigned": np.zeros((n, 2), dtype=np.uint8), "game_id": np.zeros(n, dtype=np.uint32),
           "enable_resign": np.zeros(n, dtype=np.uint8), "final_black": np.zeros(n, dtype=np.uint64),
           "final_white": np.zeros(n, dtype=np.uint64)}
    for g, (plies, s) in enumerate(games):
        raw["n_plies"][g] = len(plies)
        raw["status"][g] = s["winner"]
        raw["resigned"][g] = [s.get("resigned_black", 0), s.get("resigned_white", 0)]
        raw["game_id"][g] = s.get("game_id", g)
        raw["enable_resign"][g] = s.get("enable_resign", 1)
        for j, p in enumerate(plies):
            h = raw["headers"][g, j]
            h["own"], h["enemy"], h["n"], h["q"] = p["own"], p["enemy"], p["n"], p["q"]
            h["action"], h["player"], h["turn"], h["has_row"] = p["action"], p["player"], p.get("turn", 0), int(p["has_row"])
            h["flags"] = int(bool(p.get("solved", False)))
            raw["root_n"][g, j] = np.asarray(p["root_n"], dtype=np.float64).astype(np.uint32)
    return raw


def test_float_repr_matches_python():
    """raz_format_float_repr == float.__repr__ (what json.dump writes): visit-count ratios, random bit
    patterns, subnormals, the exponent / fixed notation switch points."""
    import ctypes
    import math
    import random
    import struct
    from reversi_alpha_zero_amd._native import lib
    buf = ctypes.create_string_buffer(32)
    rng = random.Random(3)
    xs = [0.0, 1.0, 0.1, 1e-5, 0.0001, 1e16, 1e15, 123456789012345678.0, 1e22, 1.5e-7, 2 / 3, 0.1 + 0.2, 5e-324,
          1.7976931348623157e308, -0.0, -2.5, 9999999999999998.0, 0.30000000000000004]
    xs += [rng.randint(0, 3200) / rng.randint(1, 200000) for _ in range(20000)]
    xs += [struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0] for _ in range(20000)]
    xs += [struct.unpack("<d", struct.pack("<Q", rng.getrandbits(52)))[0] for _ in range(2000)]
    for x in xs:
        if math.isnan(x) or math.isinf(x):
            continue
        assert lib.raz_format_float_repr(x, buf) == len(repr(x)) and buf.value.decode() == repr(x), repr(x)


def test_native_rows_json_equals_reference_play_files():
    """raz_emit_game_rows_json on engine-style records == json.dumps(rows) byte for byte: the sha256 of the files the
    unmodified reference worker wrote (all golden sets: tau-1 and one-hot saved policies, resignations, solver moves
    without rows, draws), and == the Python row builder on every game."""
    from reversi_alpha_zero_amd.worker.self_play import rows_of_game, game_rows_json
    from oracle_util import load_par_golden
    n_files = 0
    for gold in (load_mcts_golden(), load_par_golden(), load_mcts_golden("mcts_series_games.json")):
        for g in gold["games"]:
            plies = _plies_of(g)
            raw = _raw_of_games([(plies, {"winner": g["winner"]})])
            ctt = g["resolved_play"]["change_tau_turn"]
            tau1 = g["resolved_play_data"]["save_policy_of_tau_1"]
            for j, p in enumerate(plies):   # the golden plies do not store the turn: recover it like the env does
                raw["headers"][0, j]["turn"] = bin(p["own"]).count("1") + bin(p["enemy"]).count("1") - 4
            text, nrows = game_rows_json(raw["headers"][0], raw["root_n"][0], raw["n_plies"][0], g["winner"], ctt, tau1)
            rows = rows_of_game(plies, g["winner"])
            assert nrows == len(rows) and "[" + text + "]" == json.dumps(rows), g["variant"]
            if g["play_rows_sha256"] is not None:
                assert hashlib.sha256(("[" + text + "]").encode()).hexdigest() == g["play_rows_sha256"], g["variant"]
                n_files += 1
    assert n_files >= 20


def test_emit_raw_writes_the_same_files_as_emit(tmp_path):
    """BatchedSelfPlayWorker.emit_raw (native text, raw arrays) and .emit (Python rows, json.dump) produce identical
    play_*.json contents and GGF files for the same games, incl. file grouping, dropped draws and resign bookkeeping."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.engine import saved_policy
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    rng = np.random.default_rng(12)
    games = []
    for i in range(11):
        plies = []
        for j in range(int(rng.integers(3, 12))):
            own = int(rng.integers(0, 2**63)) * 2 + 1
            enemy = int(rng.integers(0, 2**63)) & ~own
            rn = [float(v) for v in rng.integers(0, 40, 64) * (rng.random(64) < 0.3)]
            if sum(rn) == 0:
                rn[5] = 3.0
            turn = int(rng.integers(0, 40))
            plies.append({"player": 1 + j % 2, "turn": turn, "own": own, "enemy": enemy, "action": int(rng.integers(0, 64)),
                          "has_row": bool(j % 4), "solved": bool(j % 4 == 0), "sims": 10, "loops": 1,
                          "n": float(rng.integers(1, 9)), "q": float(rng.random() * 2 - 1), "root_n": rn, "root_w": None})
        games.append((plies, {"winner": int(rng.integers(1, 4)), "status": 1, "plies": len(plies), "game_id": 500 + i,
                              "enable_resign": int(i % 3 > 0), "resigned_black": int(i % 2), "resigned_white": int(i % 5 == 0)}))

    def run(kind, tau1):
        cfg = Config()
        cfg.play_data.update(dict(nb_game_in_file=3, nb_game_in_ggf_file=4, drop_draw_game_rate=0.5, save_policy_of_tau_1=tau1))
        rc = cfg.resource
        root = tmp_path / f"{kind}{int(tau1)}"
        rc.data_dir, rc.play_data_dir, rc.self_play_ggf_data_dir = str(root), str(root / "play"), str(root / "ggf")
        os.makedirs(rc.play_data_dir)
        os.makedirs(rc.self_play_ggf_data_dir)
        w = BatchedSelfPlayWorker(cfg, b"", games_in_flight=11, seed=4, device="cpu")
        if kind == "py":
            recs = [([dict(p, saved_policy=saved_policy(p["root_n"], p["turn"], cfg.play.change_tau_turn, tau1)) for p in pl], s)
                    for pl, s in games]
            w.emit(recs, first_local_idx=1)
        else:
            w.emit_raw(_raw_of_games(games), first_local_idx=1, threads=3)
        play = [open(os.path.join(rc.play_data_dir, f)).read() for f in sorted(os.listdir(rc.play_data_dir))]
        import re   # (the GGF header carries the wall-clock time: DT[...])
        ggf = [re.sub(r"DT\[[^\]]*\]", "DT[]", open(os.path.join(rc.self_play_ggf_data_dir, f)).read())
               for f in sorted(os.listdir(rc.self_play_ggf_data_dir))]
        return play, ggf, (w.resign_test_game_count, w.false_positive_count_of_resign), len(w.buffer) if kind == "py" else None

    for tau1 in (True, False):
        p_py, g_py, book_py, _ = run("py", tau1)
        p_raw, g_raw, book_raw, _ = run("raw", tau1)
        assert len(p_py) >= 3 and p_py == p_raw and g_py == g_raw and book_py == book_raw


_GATHER_RAW_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import numpy as np
import torch.distributed as dist
from test_worker_host import _fake_records, _raw_of_games
from reversi_alpha_zero_amd.worker.self_play import gather_raw
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
fix = lambda recs: [([dict(p, action=max(p["action"], 0)) for p in pl], s) for pl, s in recs]
got = gather_raw(_raw_of_games(fix(_fake_records(rank, 4))), rank, world)
if rank == 0:
    exp = _raw_of_games(fix(_fake_records(0, 4) + _fake_records(1, 4)))
    assert set(got) == set(exp) and all(np.array_equal(got[k], exp[k]) for k in exp), "gathered arrays differ"
    print("GATHER_RAW_OK", len(got["n_plies"]))
else:
    assert got is None
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_raw_two_ranks_gloo(tmp_path):
    """The collective of the worker's raw path at world_size 2 on CPU (gloo): rank-ordered concatenation."""
    script = tmp_path / "gather_raw2.py"
    script.write_text(_GATHER_RAW_SCRIPT.format(root=ROOT))
    _p = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_p)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _p, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_RAW_OK 8" in r.stdout


_GATHER_PACKED_SCRIPT = '''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from reversi_alpha_zero_amd.engine import PLY_HEADER, GAME_SUMMARY
from reversi_alpha_zero_amd.worker.self_play import gather_packed
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
def packed_of(rank, plies):   # what SelfPlayEngine.pack_records yields: dense, cut to `plies`, here as CPU tensors
    n, own = 3, 20 + 9 * rank   # rank 0 holds games of <= 20 plies, rank 1 of <= 29: the extents differ
    rng = np.random.default_rng(100 + rank)
    npl = rng.integers(5, own + 1, n)
    npl[0] = own
    plies = own if plies is None else plies
    hdr = np.zeros((n, plies), dtype=PLY_HEADER); rn = np.zeros((n, plies, 64), dtype=np.uint32); sm = np.zeros(n, dtype=GAME_SUMMARY)
    for g in range(n):
        for i in range(int(npl[g])):
            hdr[g, i]["own"] = rng.integers(0, 2**63); hdr[g, i]["action"] = rng.integers(0, 64); hdr[g, i]["n"] = float(i)
            rn[g, i] = rng.integers(0, 800, 64)
        sm[g]["game_id"] = 10 * rank + g; sm[g]["n_plies"] = npl[g]; sm[g]["status"] = 1 + g % 3
        sm[g]["final_black"] = rng.integers(0, 2**63); sm[g]["resigned_white"] = g & 1; sm[g]["enable_resign"] = 1
    return {{"headers": torch.from_numpy(hdr.view(np.uint8).reshape(n, plies, 48)), "root_n": torch.from_numpy(rn.view(np.int32)),
             "summary": torch.from_numpy(sm.view(np.uint8).reshape(n, 32))}}
raw, moved = gather_packed(lambda plies: packed_of(rank, plies), rank, world)
if rank == 0:
    assert raw["headers"].shape == (6, 29) and raw["root_n"].shape == (6, 29, 64) and raw["root_n"].dtype == np.uint32
    assert list(raw["game_id"]) == [0, 1, 2, 10, 11, 12] and list(raw["n_plies"][[0, 3]]) == [20, 29]
    for r in range(2):
        exp = packed_of(r, 29)
        eh = exp["headers"].numpy().view(PLY_HEADER).reshape(3, 29)
        assert np.array_equal(raw["headers"][3 * r:3 * r + 3], eh) and np.array_equal(raw["root_n"][3 * r:3 * r + 3].view(np.int32), exp["root_n"].numpy())
        es = exp["summary"].numpy().view(GAME_SUMMARY).reshape(-1)
        assert list(raw["status"][3 * r:3 * r + 3]) == list(es["status"]) and list(raw["resigned"][3 * r:3 * r + 3, 1]) == list(es["resigned_white"])
        assert list(raw["final_black"][3 * r:3 * r + 3]) == list(es["final_black"])
    print("GATHER_PACKED_OK", len(raw["n_plies"]), moved)
else:
    assert raw is None
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_packed_two_ranks_gloo(tmp_path):
    """The worker's record gather on packed (dense, ply-trimmed) arrays at world_size 2 on CPU (gloo): the ranks'
    extents differ, one all_reduce(MAX) aligns them, rank-ordered concatenation, summaries decoded."""
    script = tmp_path / "gather_packed2.py"
    script.write_text(_GATHER_PACKED_SCRIPT.format(root=ROOT))
    _p = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_p)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _p, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_PACKED_OK 6" in r.stdout


def _model_config(tmp_path):
    from reversi_alpha_zero_amd.config import Config
    cfg = Config()
    rc = cfg.resource
    rc.model_dir = str(tmp_path / "model")
    rc.model_best_config_path = os.path.join(rc.model_dir, "model_best_config.json")
    rc.model_best_weight_path = os.path.join(rc.model_dir, "model_best_weight.h5")
    rc.next_generation_model_dir = os.path.join(rc.model_dir, "next_generation")
    os.makedirs(rc.next_generation_model_dir)
    cfg.model.update(dict(cnn_filter_num=16, res_layer_num=1, value_fc_size=16))
    return cfg


def test_keras_named_npz_bridge_round_trip(tmp_path):
    """SURVEY 8(f)3 weight interchange: torch module -> Keras-named arrays (Keras layouts) -> .npz under the
    reference's file name -> module -> blob, bit-exact; layer names with an arbitrary creation-number offset (a second
    model built in one Keras session) still load; a torch state-dict file still loads; a damaged HDF5 file is refused
    (real ones: tests/test_keras_h5.py)."""
    import re
    import torch
    from reversi_alpha_zero_amd.agent.model import (ReversiNet, ReversiModel, keras_named_arrays, net_from_keras_named_arrays)
    net = ReversiNet(32, 2, 48).keras_init_(1).randomize_bn_(2)
    arrs = keras_named_arrays(net)
    assert arrs["conv2d_1/kernel:0"].shape == (3, 3, 2, 32) and arrs["conv2d_6/kernel:0"].shape == (1, 1, 32, 2)
    assert arrs["policy_out/kernel:0"].shape == (128, 64) and arrs["dense_1/kernel:0"].shape == (64, 48) and arrs["value_out/kernel:0"].shape == (48, 1)
    shifted = {}
    for k, v in arrs.items():
        lname, _, w = k.partition("/")
        m = re.search(r"_(\d+)$", lname)
        if m and lname.startswith(("conv2d", "batch_normalization", "dense")):
            lname = f"{lname[:m.start()]}_{int(m.group(1)) + 37}"
        shifted[f"{lname}/{w}"] = v
    assert net_from_keras_named_arrays(shifted).to_blob() == net.to_blob()
    cfg = _model_config(tmp_path)
    cpath, wpath = str(tmp_path / "c.json"), str(tmp_path / "model_weight.h5")
    m = ReversiModel(cfg)
    m.model = net
    m.save(cpath, wpath, weight_format="npz")
    with np.load(wpath) as z:
        assert set(z.files) == set(arrs)
    m2 = ReversiModel(cfg)
    assert m2.load(cpath, wpath) and m2.model.to_blob() == net.to_blob() and m2.digest == m.digest
    torch.save(net.state_dict(), str(tmp_path / "legacy.h5"))
    m3 = ReversiModel(cfg)
    assert m3.load(cpath, str(tmp_path / "legacy.h5")) and m3.model.to_blob() == net.to_blob()
    (tmp_path / "keras.h5").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(ValueError, match="offset/length|truncated"):
        ReversiModel(cfg).load(cpath, str(tmp_path / "keras.h5"))
    with pytest.raises(ValueError):
        net_from_keras_named_arrays({k: v for k, v in arrs.items() if not k.startswith("dense_1")})


def test_load_model_and_reload_semantics(tmp_path):
    """agent/api.py:102-125 as the worker's start() applies it: nothing on disk -> fresh net saved as the best model;
    newest next-generation model preferred when play.use_newest_next_generation_model (else the best model); opts.new
    skips loading; try_reload_model picks up a changed file by digest and reports no change otherwise."""
    from reversi_alpha_zero_amd.agent.model import ReversiModel
    from reversi_alpha_zero_amd.worker.self_play import load_model, try_reload_model
    cfg = _model_config(tmp_path)
    rc = cfg.resource
    m = load_model(cfg)
    assert os.path.exists(rc.model_best_weight_path) and m.digest == ReversiModel.fetch_digest(rc.model_best_weight_path)
    best_blob = m.model.to_blob()
    assert try_reload_model(cfg, m) is False

    def add_next_gen(name, seed):
        d = os.path.join(rc.next_generation_model_dir, rc.next_generation_model_dirname_tmpl % name)
        os.makedirs(d)
        ng = ReversiModel(cfg)
        ng.build(seed=seed)
        ng.save(os.path.join(d, rc.next_generation_model_config_filename), os.path.join(d, rc.next_generation_model_weight_filename))
        return ng.model.to_blob()
    b1 = add_next_gen("20260101-000000.000000", 11)
    assert cfg.play.use_newest_next_generation_model is True
    assert try_reload_model(cfg, m) is True and m.model.to_blob() == b1 != best_blob
    assert try_reload_model(cfg, m) is False
    b2 = add_next_gen("20260102-000000.000000", 12)
    assert load_model(cfg).model.to_blob() == b2           # newest directory wins
    cfg.play.use_newest_next_generation_model = False
    assert load_model(cfg).model.to_blob() == best_blob
    m_best = load_model(cfg)
    assert try_reload_model(cfg, m_best) is False
    cfg.opts = types.SimpleNamespace(new=True)
    fresh = load_model(cfg)                                  # opts.new: build + overwrite the best model
    assert fresh.digest == ReversiModel.fetch_digest(rc.model_best_weight_path)
    other = ReversiModel(cfg)
    other.build(seed=5)
    other.save(rc.model_best_config_path, rc.model_best_weight_path)   # the eval worker promotes a new best model ...
    assert try_reload_model(cfg, m_best) is True and m_best.model.to_blob() == other.model.to_blob()   # ... a running worker picks it up


def test_threshold_update_once_per_emitted_batch(tmp_path):
    """worker/self_play.py:250-260 under batching: 1000 games finished under one threshold step it ONCE (by the rate over
    the batch's no-resign test games), not once per 100 test games."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    cfg = Config()
    w = BatchedSelfPlayWorker(cfg, b"", games_in_flight=8)
    w._defer_threshold_update = True
    t0 = cfg.play.resign_threshold
    for i in range(1000):   # 500 test games, 40 % false positives (>= false_positive_threshold 0.05)
        w.finish_game({"winner": 1, "resigned_black": int(i % 5 < 2), "resigned_white": 0, "enable_resign": i % 2})
    assert cfg.play.resign_threshold == t0 and w.resign_test_game_count == 500
    w._defer_threshold_update = False
    w.check_and_update_resignation_threshold()
    assert abs(cfg.play.resign_threshold - (t0 - cfg.play.resign_threshold_delta)) < 1e-12 and w.resign_test_game_count == 0


def test_the_fused_kernel_is_chosen_where_it_applies_and_pays():
    """worker/self_play.py wants_fused_tree_net: 16-filter nets on the default net kernels; "auto" not with the evaluation cache and not with
    the end-game solver on (round 5: the two-kernel pipeline is the faster one once solves go through the pool)."""
    import types
    from reversi_alpha_zero_amd.worker.self_play import wants_fused_tree_net
    off = types.SimpleNamespace(use_solver_turn=0, use_solver_turn_in_simulation=0)
    on = types.SimpleNamespace(use_solver_turn=50, use_solver_turn_in_simulation=50)
    insim = types.SimpleNamespace(use_solver_turn=0, use_solver_turn_in_simulation=50)
    assert wants_fused_tree_net("auto", 16, 16, 0, None, off)
    assert not wants_fused_tree_net("auto", 16, 16, 0, None, on) and not wants_fused_tree_net("auto", 16, 16, 0, None, insim)
    assert not wants_fused_tree_net("auto", 16, 16, 0, 20, off)            # evaluation cache attached
    assert wants_fused_tree_net(True, 16, 16, 0, 20, on)                   # forced: wherever it applies
    assert not wants_fused_tree_net(False, 16, 16, 0, None, off)
    assert not wants_fused_tree_net(True, 32, 16, 0, None, off) and not wants_fused_tree_net(True, 16, 2048, 0, None, off)
    assert not wants_fused_tree_net(True, 16, 16, 1, None, off)            # a forced net kernel (tests)
