"""GPU: continuous batching (raz_engine_harvest / SelfPlayEngine.play_continuous; the reference worker starts its next
game the moment one ends, worker/self_play.py:95-137).  Slots are refilled with the next unplayed game id as games
finish; the results must be the games the oracle plays for those ids - whatever the batch size, the slot a game lands in,
or the order games finish in - and the id-ordered outbox must equal what lock-step batches of the same ids record."""
import numpy as np
import pytest

import oracle as O
from oracle_util import load_mcts_golden, load_par_golden, golden_net_blob, config_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold():
    return load_mcts_golden()


@pytest.fixture(scope="module")
def blob(gold):
    return golden_net_blob(gold["net"])


def _raw(outbox):
    from reversi_alpha_zero_amd.engine import raw_from_packed
    assert bool(outbox["done"].all())
    return raw_from_packed(*(outbox[k].cpu().numpy() for k in ("headers", "root_n", "summary")))


def _check_vs_oracle(raw, cfg, blob, seed, first, ids, sims_of, thr_of=None, par=1):
    for gid in ids:
        c = cfg
        if thr_of is not None:
            import copy
            c = copy.deepcopy(cfg)
            c.play.resign_threshold = thr_of(gid)
        plies, summ = O.selfplay_game(O.play_cfg_from_config(c, parallel_search_num=par), blob, seed, gid, sims_of(gid))
        r = gid - first
        assert int(raw["game_id"][r]) == gid
        n = int(raw["n_plies"][r])
        assert n == len(plies), (gid, n, len(plies))
        assert [int(a) for a in raw["headers"][r, :n]["action"]] == [p["action"] for p in plies], gid
        for i, p in enumerate(plies):
            assert [float(x) for x in raw["root_n"][r, i]] == p["root_n"], (gid, i)
            assert bool(raw["headers"][r, i]["has_row"]) == p["has_row"] and float(raw["headers"][r, i]["q"]) == (p["q"] if p["action"] >= 0 else 0.0)
        assert int(raw["status"][r]) & 0x0f == summ["winner"] and (int(raw["final_black"][r]), int(raw["final_white"][r])) == (summ["black"], summ["white"])
        assert (int(raw["resigned"][r, 0]), int(raw["resigned"][r, 1]), int(raw["enable_resign"][r])) == \
            (summ["resigned_black"], summ["resigned_white"], summ["enable_resign"])


@pytest.mark.parametrize("slots", [8, 13])
def test_continuous_batching_equals_oracle_whatever_the_batch_size(gold, blob, slots):
    """50 game ids through 8 and through 13 slots (games of very different lengths: resignation on, id-dependent
    simulations per move, the resign threshold changing at id 130): id-ordered outbox == oracle games; totals add up."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in gold["games"] if g["variant"] == "agz_resign")
    cfg = config_of(g0)
    first, total = 100, 50
    sims_of = lambda gid: 8 + (gid % 4) * 5
    thr_of = lambda gid: -0.02 if gid < 130 else -0.3
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=slots, seed=17, sims_hint=23)
    outbox, st = eng.play_continuous(first, total, sims_of, chunk=48, resign_threshold_of=thr_of)
    raw = _raw(outbox)
    assert list(raw["game_id"]) == list(range(first, first + total))
    _check_vs_oracle(raw, cfg, blob, 17, first, list(range(first, first + total, 3)) + [129, 130, 131], sims_of, thr_of)
    assert st["finished_games"] == total
    assert st["total_sims"] == int(raw["headers"]["sims"].sum())
    lens = raw["n_plies"]
    assert lens.min() < 0.7 * lens.max(), "the games should differ in length for this test to mean anything"


def test_continuous_equals_lock_step_batches_and_keeps_the_slots_busy(gold, blob):
    """The same 64 ids as 4 lock-step batches of 16 and as one continuous run on 16 slots: identical records; the
    continuous run needs fewer steps (finished slots do not idle until the slowest game of the batch ends) and
    keeps >= 85 % of the leaf slots busy even with 16 slots draining at the end."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in gold["games"] if g["variant"] == "agz_resign")
    cfg = config_of(g0)
    dnet = DeviceNet(blob, DEV)
    eng = SelfPlayEngine(cfg, dnet, n_games=16, seed=5, sims_hint=12)
    lock, lock_steps = [], 0
    for b in range(4):
        eng.start(200 + 16 * b, 12)
        lock_steps += eng.run(chunk=16)["steps"]
        lock += eng.records()
    outbox, st = eng.play_continuous(200, 64, lambda gid: 12, chunk=16)
    raw = _raw(outbox)
    for r, (plies, summ) in enumerate(lock):
        n = len(plies)
        assert int(raw["n_plies"][r]) == n and int(raw["game_id"][r]) == summ["game_id"] == 200 + r
        assert [int(a) for a in raw["headers"][r, :n]["action"]] == [p["action"] for p in plies]
        assert all([float(x) for x in raw["root_n"][r, i]] == plies[i]["root_n"] for i in range(n))
        assert int(raw["status"][r]) == summ["status"]
    print(f"continuous: {st['steps']} steps, occupancy {st['leaf_slot_occupancy']:.3f}; lock-step: {lock_steps} steps")
    assert st["steps"] < lock_steps


def test_continuous_batching_with_simulation_slots_solver_and_pruning(blob):
    """mini.yml as shipped but for the per-game tree reset (parallel_search_num 4: the slot kernel, thinking_loop 2, solver
    from turn 50) with node pools far too small for a game (k_gc between harvests): 20 ids on 6 slots == oracle."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    par = load_par_golden()
    g0 = next(g for g in par["games"] if g["variant"] == "mini_par4_as_shipped")
    cfg = config_of(g0)
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=6, seed=23, nodes_per_game=1536)
    outbox, st = eng.play_continuous(300, 20, lambda gid: 14, chunk=24)
    assert st["gc_runs"] >= 1
    _check_vs_oracle(_raw(outbox), cfg, blob, 23, 300, range(300, 320, 2), lambda gid: 14, par=4)


def test_worker_files_do_not_depend_on_slots_or_refill(gold, blob, tmp_path):
    """The `self` worker: 36 game ids as one block on 6 slots (continuous batching: each slot plays ~6 games) and as one
    lock-step batch of 36 slots -> byte-identical play_*.json files (games are emitted in id order either way); and the same with
    the block's finished prefix handed to the file writer while the block is still played (streamed emission)."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    g0 = next(g for g in gold["games"] if g["variant"] == "agz_resign")
    outs = []
    for tag, slots, block in (("refill", 6, 36), ("lockstep", 36, 36), ("refill_streamed", 6, 36)):
        cfg = Config()
        cfg.play.update(g0["resolved_play"])
        cfg.play.schedule_of_simulation_num_per_move = [(0, 9), (20, 14)]
        cfg.play_data.update(dict(g0["resolved_play_data"], nb_game_in_file=7, enable_ggf_data=False))
        rc = cfg.resource
        out = tmp_path / tag
        rc.data_dir = str(out); rc.play_data_dir = str(out / "play_data"); rc.self_play_ggf_data_dir = str(out / "ggf")
        rc.model_dir = str(out / "model"); rc.next_generation_model_dir = str(out / "model" / "next"); rc.log_dir = str(out / "logs")
        rc.project_dir = str(out); rc.force_simulation_num_file = str(out / ".force-sim"); rc.self_play_game_idx_file = str(out / ".idx")
        w = BatchedSelfPlayWorker(cfg, blob, games_in_flight=slots, seed=8, device=DEV, block_games=block)
        if tag == "refill_streamed":   # streamed emission (run()): the finished prefix of the block goes to the writer in pieces of one file
            w.stream_piece_games = 7
        w.run(total_games=36)
        assert (getattr(w, "_streamed_rows", 0) >= 14) == (tag == "refill_streamed"), (tag, getattr(w, "_streamed_rows", None))
        files = sorted((out / "play_data").iterdir())
        outs.append([f.read_text() for f in files])
        assert (out / ".idx").read_text() == "36"
        if tag == "refill":
            assert w.last_stats["steps"] > 0 and w.last_stats["finished_games"] == 36
    assert len(outs[0]) == 5 and outs[0] == outs[1] == outs[2]
