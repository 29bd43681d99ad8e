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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
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
