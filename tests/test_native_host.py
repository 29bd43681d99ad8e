"""CPU: the host side of libraz — C-ABI surface, scalar primitives, and the ReversiEnv / bitboard
facade — against the golden vectors.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

from conftest import H, ROOT


def test_library_exports_every_declared_symbol():
    from reversi_alpha_zero_amd import _native
    hdr = open(os.path.join(ROOT, "include", "raz.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(raz_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    for name in sorted(declared):
        assert hasattr(_native.lib, name), f"{name} declared in include/raz.h but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature in _native.py"
    assert _native.lib.raz_abi_version() == 3


def test_scalar_primitives_vs_golden(golden_bb):
    from reversi_alpha_zero_amd.lib import bitboard as bb
    for rec in golden_bb["positions"] + [tb[s] for tb in golden_bb["test_boards"]
                                         for s in ("black_to_move", "white_to_move")]:
        own, enemy = H(rec["own"]), H(rec["enemy"])
        assert bb.find_correct_moves(own, enemy) == H(rec["legal"])
        for a, f in rec["flips"].items():
            assert bb.calc_flip(int(a), own, enemy) == H(f)
    for rec in golden_bb["garbage"]:
        own, enemy = H(rec["own"]), H(rec["enemy"])
        assert bb.find_correct_moves(own, enemy) == H(rec["legal"])
        assert bb.calc_flip(rec["pos"], own, enemy) == H(rec["flip"])
    for rec in golden_bb["symmetries"]:
        x = H(rec["x"])
        for name in ("flip_vertical", "flip_diag_a1h8", "rotate90", "rotate180"):
            assert getattr(bb, name)(x) == H(rec[name])
        assert bb.bit_count(x) == rec["bit_count"]
        arr = bb.bit_to_array(x, 64)
        assert arr.dtype == np.uint8 and "".join(str(v) for v in arr) == rec["bit_to_array"]


def test_calc_flip_range_assert():
    from reversi_alpha_zero_amd.lib import bitboard as bb
    from reversi_alpha_zero_amd import _native
    with pytest.raises(AssertionError):
        bb.calc_flip(64, 1, 2)
    assert _native.lib.raz_calc_flip(64, 1, 2) == 0
    assert "out of range" in _native.last_error()


def test_env_facade_playouts(golden_bb):
    from reversi_alpha_zero_amd.env.reversi_env import ReversiEnv, Player, Winner
    for g in golden_bb["playouts"]:
        env = ReversiEnv().reset()
        for a, p in zip(g["actions"], g["players"]):
            assert not env.done and env.next_player.value == p
            board, info = env.step(a)
            assert board is env.board and info == {}
        assert env.done and env.winner == Winner(g["winner"]) and env.turn == g["turn"]
        assert (env.board.black, env.board.white) == (H(g["black"]), H(g["white"]))


def test_env_facade_edges(golden_bb):
    from reversi_alpha_zero_amd.env.reversi_env import ReversiEnv, Player, Winner
    for rec in golden_bb["env_edge"]:
        if rec["desc"] == "update_zero_boards":
            env = ReversiEnv().update(0, 0, Player.white)
            assert (env.board.black, env.board.white, env.turn) == (H(rec["black"]), H(rec["white"]), rec["turn"])
            continue
        env = ReversiEnv().reset()
        env.next_player = Player(rec["player_in"])
        env.step(None if rec["action"] < 0 else rec["action"])
        assert (env.board.black, env.board.white) == (H(rec["black"]), H(rec["white"]))
        assert (env.next_player.value, env.turn, env.done, env.winner.value if env.winner else 0) == \
            (rec["next_player"], rec["turn"], rec["done"], rec["winner"])
    with pytest.raises(AssertionError):
        ReversiEnv().reset().step(64)


def test_batched_ops_refuse_cpu_tensors():
    import torch
    from reversi_alpha_zero_amd.lib import bitboard as bb
    t = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(ValueError):
        bb.legal_moves_batch(t, t)


def test_engine_workspace_scales_with_parallel_search_num():
    """Host-only part of the engine ABI: raz_engine_workspace_bytes validates the config and sizes the
    per-slot arrays (simulation slots, paths, leaf-exchange rows) by parallel_search_num (config.py:142)."""
    import ctypes
    import types
    from reversi_alpha_zero_amd import _native as N
    from reversi_alpha_zero_amd.engine import engine_config_from
    play = types.SimpleNamespace(
        share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=40,
        start_rethinking_turn=10, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=10,
        virtual_loss=3, parallel_search_num=1, resign_threshold=-0.9, allowed_resign_turn=10,
        disable_resignation_rate=0.1, use_solver_turn=0, use_solver_turn_in_simulation=0)
    cfg = types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
    sizes = {}
    for k in (1, 4, 16):
        play.parallel_search_num = k
        c = engine_config_from(cfg, n_games=64, seed=0, nodes_per_game=256)
        assert c.parallel_search_num == k
        sizes[k] = N.lib.raz_engine_workspace_bytes(ctypes.byref(c))
        assert sizes[k] > 0
    per_slot = 256 + 64 * 9 + 1 + 16 + 260     # slot block, path (node, mirror, action), flag, leaf in, leaf out
    assert sizes[4] - sizes[1] >= 64 * (3 * per_slot)                  # the slot kernel's arrays for 4 slots ...
    assert sizes[16] - sizes[4] >= 64 * (12 * per_slot)                # ... and 12 more
    play.parallel_search_num = 17                                      # prediction_queue_size (config.py:141)
    with pytest.raises(ValueError):
        engine_config_from(cfg, n_games=64, seed=0, nodes_per_game=256)
    c = engine_config_from(types.SimpleNamespace(play=types.SimpleNamespace(**dict(vars(play), parallel_search_num=16)),
                                                 play_data=cfg.play_data), n_games=64, seed=0, nodes_per_game=256)
    c.parallel_search_num = 17
    assert N.lib.raz_engine_workspace_bytes(ctypes.byref(c)) == 0 and b"parallel_search_num" in N.lib.raz_last_error()
    # policy_decay_turn / policy_decay_power (agent/player.py:411): inert at the shipped 60 / 3 and at anything that keeps
    # min(exp(1 - (turn / T) ** p), 1) == 1 for turn <= 60; a value that would decay the priors is refused, not ignored
    play.parallel_search_num = 1
    for turn, power, ok in ((60, 3, True), (80, 1, True), (60, 0, True), (30, 3, False), (59, 3, False), (60, -1, False)):
        play.policy_decay_turn, play.policy_decay_power = turn, power
        if ok:
            engine_config_from(cfg, n_games=64, seed=0, nodes_per_game=256)
        else:
            with pytest.raises(ValueError, match="policy_decay"):
                engine_config_from(cfg, n_games=64, seed=0, nodes_per_game=256)


def test_valu_shaped_bitboard_ops_equal_the_reference_shaped_ones(tmp_path):
    """csrc/raz_bitboard_valu.h (what the sweep kernels compute with: three-input logic, carry-propagation rows, the
    step split into first / finish) == csrc/raz_bitboard.h (pinned by the goldens and the oracle) on the host, for
    playout positions x every action and for garbage inputs: tests/native/bbv_check.cpp, built here with g++."""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "bbv_check.cpp")
    exe = str(tmp_path / "bbv_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "1500000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BBV_OK" in r.stdout, r.stdout[-2000:]


def test_bit_sliced_sweep_arithmetic_equals_the_reference_shaped_ops(tmp_path):
    """csrc/raz_sweep_sliced.h (the large-batch sweep kernels' arithmetic: 32 boards per lane, one bit per board - transposes,
    find_correct_moves as walks along the board's lines, calc_flip, the whole step with its masks) == csrc/raz_bitboard.h on the host,
    lane by lane: playout positions, overlapping / full / sparse garbage, legal moves, occupied squares, resignations, actions outside
    the board, finished games; boards with a square of both colours are REPORTED (the kernel steps those board by board):
    tests/native/sliced_check.cpp, built here with g++."""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "sliced_check.cpp")
    exe = str(tmp_path / "sliced_check")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "30000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SLICED_OK" in r.stdout, r.stdout[-2000:]


def test_integration_md_names_every_entry_point():
    """INTEGRATION.md's table of entry points covers the whole header (names may be abbreviated as `raz_engine_create/start/...`)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "raz.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(raz_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 50
    slashed = set()
    for m in re.finditer(r"`(raz_[a-z0-9_]+?)_([a-z0-9_]+(?:\s*/\s*_?[a-z0-9_]+)+)`", doc):   # `raz_engine_create/start/step` or `raz_engine_create / _start / _step`
        prefix = m.group(1)
        for part in re.split(r"\s*/\s*", m.group(2)):
            slashed.add(prefix + "_" + part.lstrip("_"))
    missing = [n for n in names if n not in doc and n not in slashed]
    assert not missing, missing
