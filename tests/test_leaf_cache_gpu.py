"""GPU: the cross-game evaluation cache (csrc/raz_leaf_cache.hip, raz_engine_set_leaf_cache).  The net is a pure,
batch-invariant function of the position, so serving a repeated position from the table must leave every game exactly as
it is without the cache - actions, root N, root W (f64 bits), resignation flags - while the net sees fewer rows."""
import types

import numpy as np
import pytest

from oracle_util import load_mcts_golden, load_par_golden, golden_net_blob, config_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold():
    return load_mcts_golden()


@pytest.fixture(scope="module")
def blob(gold):
    return golden_net_blob(gold["net"])


def _play(cfg, dnet, n, seed, sims, cache, first=0, **kw):
    from reversi_alpha_zero_amd.engine import SelfPlayEngine
    eng = SelfPlayEngine(cfg, dnet, n_games=n, seed=seed, sims_hint=sims, record_root_w=True, leaf_cache_log2=cache, **kw)
    eng.start(first, sims)
    st = eng.run(chunk=64)
    return eng.read_raw(), st, eng.leaf_cache_stats()


def _same(a, b):
    for k in ("n_plies", "status", "resigned", "final_black", "final_white"):
        assert np.array_equal(a[k], b[k]), k
    for g in range(len(a["n_plies"])):   # (record rows beyond a game's plies are never written: compare the plies played)
        n = int(a["n_plies"][g])
        assert np.array_equal(a["headers"][g, :n], b["headers"][g, :n]), g
        assert np.array_equal(a["root_n"][g, :n], b["root_n"][g, :n]), g
        assert np.array_equal(a["root_w"][g, :n].view(np.uint64), b["root_w"][g, :n].view(np.uint64)), g


@pytest.mark.parametrize("variant,n", [("mini_shared", 64), ("agz_resign", 300)])
def test_games_unchanged_by_the_cache_narrow_net(gold, blob, variant, n):
    """mini net (rows served by the cache leave the batch through the `active` mask); 300 games = 3 slices on 3 streams
    sharing one table."""
    from reversi_alpha_zero_amd.engine import DeviceNet
    g0 = next(g for g in gold["games"] if g["variant"] == variant)
    cfg = config_of(g0)
    dnet = DeviceNet(blob, DEV)
    plain, st0, c0 = _play(cfg, dnet, n, 7, 12, None)
    cached, st1, c1 = _play(cfg, dnet, n, 7, 12, 20)
    _same(plain, cached)
    assert c0 == {"hits": 0, "in_batch_duplicates": 0, "evaluated": 0, "no_room": 0}
    assert st1["nn_leaves"] == st0["nn_leaves"] == c1["hits"] + c1["in_batch_duplicates"] + c1["evaluated"]
    # (a leaf is evaluated under a random D4 symmetry, agent/player.py:300-305: a position has 8 keys, so short runs of few
    #  games share little; the openings of thousands of 800-simulation searches share a lot - tools/whole_games_config3.py)
    assert c1["hits"] > 0 and c1["in_batch_duplicates"] > 0 and c1["no_room"] == 0, c1
    print(variant, c1, "of", st1["nn_leaves"], "leaves")


@pytest.mark.parametrize("par", [1, 4])
def test_games_unchanged_by_the_cache_wide_net_compacted(par):
    """128-filter net on the split-f16 trunk: the rows still to evaluate are compacted (conv0 / heads go through the index
    list, the convolutions see a device-side row count).  Also with the slot kernel (4 simulations in flight per game) and a
    table far too small for the run (2^10 entries: most claims find no room and are simply evaluated)."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    blob = ReversiNet(128, 1, 64).keras_init_(7).randomize_bn_(8).to_blob()
    play = types.SimpleNamespace(
        simulation_num_per_move=14, share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=400,
        start_rethinking_turn=8, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=4, virtual_loss=3,
        parallel_search_num=par, resign_threshold=-0.9, allowed_resign_turn=50, disable_resignation_rate=0.1,
        use_solver_turn=0, use_solver_turn_in_simulation=0)
    cfg = types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
    dnet = DeviceNet(blob, DEV, kernel="f16x3")
    plain, st0, _ = _play(cfg, dnet, 40, 3, 14, None)
    cached, st1, c1 = _play(cfg, dnet, 40, 3, 14, 18)
    tiny, st2, c2 = _play(cfg, dnet, 40, 3, 14, 10)
    _same(plain, cached)
    _same(plain, tiny)
    assert c1["hits"] + c1["in_batch_duplicates"] + c1["evaluated"] == st1["nn_leaves"] == st0["nn_leaves"]
    assert c1["evaluated"] < st1["nn_leaves"] and c1["hits"] > 0 and c1["no_room"] == 0, c1
    assert c2["no_room"] > 0 and c2["evaluated"] > c1["evaluated"], c2
    assert dnet.range_ok()
    print("par", par, c1, c2, "of", st1["nn_leaves"], "leaves")


def test_continuous_batching_with_the_cache(gold, blob):
    """Refilled slots replay the openings the table already holds: 48 ids on 12 slots, outbox == the run without a cache."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in gold["games"] if g["variant"] == "agz_resign")
    cfg = config_of(g0)
    dnet = DeviceNet(blob, DEV)
    out = []
    for cache in (None, 16):
        eng = SelfPlayEngine(cfg, dnet, n_games=12, seed=11, sims_hint=10, leaf_cache_log2=cache)
        outbox, st = eng.play_continuous(500, 48, lambda gid: 10, chunk=32)
        out.append(({k: outbox[k].cpu().numpy() for k in ("headers", "root_n", "summary")}, eng.leaf_cache_stats()))
    for k in ("headers", "root_n", "summary"):
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
    assert out[1][1]["hits"] > 0
