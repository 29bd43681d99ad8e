"""GPU: the `self` worker end to end on the shape the metric is quoted on - the 256x10 net (config.py:187-193 defaults) on
the split-f16 trunk, ch5.yml play settings, 512 slots with continuous batching (2 games per slot per block), node pools
capped far below a whole game's tree so that k_gc runs between harvests - at a reduced number of simulations per move so
that it finishes in seconds.  The files must hold exactly the rows of the games the oracle plays for those ids when it is
fed with the device net's outputs (the reference's NN seam), and the worker must report the block's throughput."""
import json
import types

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_worker_block_on_the_metric_shape(tmp_path):
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    from reversi_alpha_zero_amd.lib.data_helper import get_game_data_filenames, read_game_data_from_file
    from oracle_util import rows_of_game
    cfg = Config()
    cfg.play.update(dict(thinking_loop=1, c_puct=5, allowed_resign_turn=50, use_solver_turn=0, use_solver_turn_in_simulation=0,
                         parallel_search_num=1, schedule_of_simulation_num_per_move=[(0, 12)]))   # ch5.yml:9-16 + the declared overrides
    cfg.play_data.update(dict(nb_game_in_file=256, enable_ggf_data=False, drop_draw_game_rate=0.5))
    rc = cfg.resource
    rc.data_dir = str(tmp_path); rc.play_data_dir = str(tmp_path / "play_data"); rc.self_play_ggf_data_dir = str(tmp_path / "ggf")
    rc.model_dir = str(tmp_path / "model"); rc.next_generation_model_dir = str(tmp_path / "model" / "next"); rc.log_dir = str(tmp_path / "logs")
    rc.project_dir = str(tmp_path); rc.force_simulation_num_file = str(tmp_path / ".force-sim"); rc.self_play_game_idx_file = str(tmp_path / ".idx")
    blob = ReversiNet(256, 10, 256).keras_init_(0).to_blob()
    w = BatchedSelfPlayWorker(cfg, blob, games_in_flight=512, seed=2, device=DEV, block_games=1024, net_kernel="auto")
    w._pool_nodes_that_fit = lambda fraction=0.7, cache_log2=None: 320   # ~27 x sims: far below a whole game's 1744 nodes -> pruning between harvests
    w.run(total_games=1024)
    st = w.last_stats
    assert st["finished_games"] == 1024 and st["gc_runs"] >= 1 and st["leaf_slot_occupancy"] > 0.8
    assert w._net.kernel_name.startswith("f16x3") and w._net.range_ok()
    files = get_game_data_filenames(rc)
    assert len(files) == 4 and (tmp_path / ".idx").read_text() == "1024"
    got = [row for f in files for row in read_game_data_from_file(f)]
    # the oracle on sampled ids, leaves evaluated by the worker's own device net
    dnet = w._net

    def nn(own, enemy):
        to = lambda v: torch.tensor([v - (1 << 64) if v >= 1 << 63 else v], dtype=torch.int64, device=DEV)
        p, v = dnet.predict_bitboards(to(own), to(enemy))
        return p[0].cpu().numpy(), float(v[0].item())
    ocfg = O.play_cfg_from_config(cfg)
    # rows are in game-id order; locate a game's rows by replaying the oracle for ids 0..k and counting rows is too slow at
    # 1 ms per leaf, so check the first two games (a prefix of the first file) and the row count of the whole block
    exp = []
    for gid in (0, 1):
        plies, summ = O.selfplay_game(ocfg, None, 2, gid, 12, nn=nn)
        rows = rows_of_game(plies, summ["winner"])
        dropped = summ["winner"] == 3 and not (cfg.play_data.drop_draw_game_rate <= summ["drop_draw_u"])
        exp += [] if dropped else rows
    assert json.dumps(got[:len(exp)]) == json.dumps(exp)
    assert len(got) % 8 == 0 and len(got) > 1024 * 8 * 40


def _small_cfg(tmp_path, sims=6):
    from reversi_alpha_zero_amd.config import Config
    cfg = Config()
    cfg.play.update(dict(thinking_loop=1, c_puct=5, allowed_resign_turn=50, use_solver_turn=0, use_solver_turn_in_simulation=0,
                         parallel_search_num=1, schedule_of_simulation_num_per_move=[(0, sims)]))
    cfg.play_data.update(dict(nb_game_in_file=64, enable_ggf_data=False, drop_draw_game_rate=0.0))
    rc = cfg.resource
    rc.data_dir = str(tmp_path); rc.play_data_dir = str(tmp_path / "play_data"); rc.self_play_ggf_data_dir = str(tmp_path / "ggf")
    rc.model_dir = str(tmp_path / "model"); rc.next_generation_model_dir = str(tmp_path / "model" / "next"); rc.log_dir = str(tmp_path / "logs")
    rc.project_dir = str(tmp_path); rc.force_simulation_num_file = str(tmp_path / ".force-sim"); rc.self_play_game_idx_file = str(tmp_path / ".idx")
    return cfg


def _file_bytes(cfg):
    from reversi_alpha_zero_amd.lib.data_helper import get_game_data_filenames
    return [open(f, "rb").read() for f in get_game_data_filenames(cfg.resource)]


def test_worker_replays_a_block_on_f32_when_the_split_trunk_overflows(tmp_path):
    """A net whose activations leave the f16 range (raz_net_range_check): run() must not raise - it plays the block again on
    the exact-f32 kernels (include/raz.h raz_net.reserved = 0) and its files are byte for byte those of a worker that ran
    on the exact-f32 kernels from the start."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    net = ReversiNet(128, 1, 64).keras_init_(5)
    with torch.no_grad():
        net.stem.conv.weight.mul_(1.0e6)
    blob = net.to_blob()
    out = {}
    for kernel in ("auto", "f32"):
        cfg = _small_cfg(tmp_path / kernel)
        w = BatchedSelfPlayWorker(cfg, blob, games_in_flight=64, seed=3, device=DEV, block_games=128, net_kernel=kernel, leaf_cache_log2=None)
        w.run(total_games=128)
        assert w._net.kernel_name == "f32" and w._f32_fallback == (kernel == "auto")
        out[kernel] = _file_bytes(cfg)
    assert len(out["auto"]) == 2 and out["auto"] == out["f32"]


def test_worker_reloads_weights_between_blocks_with_capped_pools(tmp_path):
    """agent/api.py:117-125 try_reload_model between two blocks, on pools sized from the free device memory: the old engine's
    workspace must be released before the new one is sized (the caller's `packed` closure, the cached allocator blocks), so
    the second engine gets pools as large as the first one's and the device's reserved memory does not double; the second
    block's files are those a fresh worker writes for the same ids with the new weights."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    blobs = [ReversiNet(128, 1, 64).keras_init_(s).to_blob() for s in (1, 2)]
    cfg = _small_cfg(tmp_path / "reload")
    torch.cuda.empty_cache()
    w = BatchedSelfPlayWorker(cfg, blobs[0], games_in_flight=256, seed=3, device=DEV, block_games=512, leaf_cache_log2=12)
    sized, reserved = [], []
    real = w._pool_nodes_that_fit

    def capped(fraction=0.7, cache_log2=None):
        cap = real(fraction, cache_log2)          # what 70 % of the FREE memory holds: halves if the old engine is still alive
        sized.append(cap)
        reserved.append(torch.cuda.memory_reserved())
        return 320                                # (the pools the test really uses: pruned between harvests)
    w._pool_nodes_that_fit = capped
    polls = []

    def reload_model():
        polls.append(1)
        return blobs[1] if len(polls) == 1 else None
    w.run(total_games=1024, reload_model=reload_model)
    assert len(sized) == 2 and sized[1] >= 0.95 * sized[0], sized
    assert reserved[1] <= reserved[0] + (64 << 20), reserved     # nothing of the first engine was still reserved
    second = _file_bytes(cfg)[512 // 64:]
    cfg2 = _small_cfg(tmp_path / "fresh")
    (tmp_path / "fresh").mkdir(exist_ok=True)
    (tmp_path / "fresh" / ".idx").write_text("512")
    w2 = BatchedSelfPlayWorker(cfg2, blobs[1], games_in_flight=256, seed=3, device=DEV, block_games=512, leaf_cache_log2=12)
    w2._pool_nodes_that_fit = lambda fraction=0.7, cache_log2=None: 320
    w2.run(total_games=512)
    assert second == _file_bytes(cfg2) and len(second) == 8
