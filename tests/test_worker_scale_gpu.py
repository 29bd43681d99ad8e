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
    w._pool_nodes_that_fit = lambda fraction=0.7: 320   # ~27 x sims: far below a whole game's 1744 nodes -> pruning between harvests
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
