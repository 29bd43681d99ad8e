"""GPU: the HIP net kernel and the batched MCTS self-play engine (through the C ABI) against the CPU
oracle — bit-exact — and against the committed golden games played by the unmodified reference."""
import numpy as np
import pytest
import torch

import oracle as O
from oracle_util import load_mcts_golden, golden_net_blob, config_of, dense  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def golden():
    return load_mcts_golden()


@pytest.fixture(scope="module")
def blob(golden):
    return golden_net_blob(golden["net"])


def _positions(n, seed):
    rng = np.random.default_rng(seed)
    own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    enemy = rng.integers(0, 2**64, size=n, dtype=np.uint64) & ~own
    return own, enemy


@pytest.mark.parametrize("shape", [(16, 1, 16), (32, 2, 48), (64, 3, 80)])
def test_net_mfma_kernel_equals_valu_kernel(shape):
    """k_net_mfma (matrix cores) and k_net_wave (VALU) are the same function bit for bit."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    blob = ReversiNet(*shape).keras_init_(2).randomize_bn_(3).to_blob()
    own, enemy = _positions(777, 4)
    o, e = torch.from_numpy(own.view(np.int64)).to(DEV), torch.from_numpy(enemy.view(np.int64)).to(DEV)
    pa, va = DeviceNet(blob, DEV).predict_bitboards(o, e)
    pb, vb = DeviceNet(blob, DEV, force_valu_kernel=True).predict_bitboards(o, e)
    assert torch.equal(pa.view(torch.int32), pb.view(torch.int32))
    assert torch.equal(va.view(torch.int32), vb.view(torch.int32))


@pytest.mark.parametrize("shape,n", [((128, 1, 64), 37), ((256, 2, 256), 21)])
def test_net_wide_kernel_equals_valu_kernel(shape, n):
    """k_conv3x3_wide (implicit GEMM on v_mfma_f32_32x32x2) == k_net_wave (VALU) bit for bit, ragged n."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    blob = ReversiNet(*shape).keras_init_(8).randomize_bn_(9).to_blob()
    own, enemy = _positions(n, 5)
    o, e = torch.from_numpy(own.view(np.int64)).to(DEV), torch.from_numpy(enemy.view(np.int64)).to(DEV)
    pa, va = DeviceNet(blob, DEV).predict_bitboards(o, e)
    pb, vb = DeviceNet(blob, DEV, force_valu_kernel=True).predict_bitboards(o, e)
    assert torch.equal(pa.view(torch.int32), pb.view(torch.int32))
    assert torch.equal(va.view(torch.int32), vb.view(torch.int32))


@pytest.mark.parametrize("shape,n", [((16, 1, 16), 300), ((32, 2, 48), 40), ((64, 1, 32), 10), ((128, 1, 256), 3),
                                     ((256, 1, 64), 2)])
def test_net_kernel_bitwise_vs_oracle_and_torch(shape, n):
    """HIP forward == oracle forward bit for bit; both within 1e-5 of the fp32 torch graph."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    net = ReversiNet(*shape).keras_init_(5).randomize_bn_(6)
    blob = net.to_blob()
    dnet = DeviceNet(blob, DEV)
    own, enemy = _positions(n, 1)
    pol, val = dnet.predict_bitboards(torch.from_numpy(own.view(np.int64)).to(DEV), torch.from_numpy(enemy.view(np.int64)).to(DEV))
    pol, val = pol.cpu().numpy(), val.cpu().numpy()
    lib = O.load_ext()
    for i in range(n):
        p = np.zeros(64, np.float32)
        v = np.zeros(1, np.float32)
        assert lib.orc_net_forward(blob, len(blob), int(own[i]), int(enemy[i]), p.ctypes.data, v.ctypes.data) == 0
        assert np.array_equal(p.view(np.uint32), pol[i].view(np.uint32)), i
        assert v.view(np.uint32)[0] == val[i:i + 1].view(np.uint32)[0], i
    bits = lambda a: ((a[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float32)
    x = torch.from_numpy(np.stack([bits(own), bits(enemy)], axis=1).reshape(n, 2, 8, 8))
    with torch.no_grad():
        tp, tv = net(x)
    assert np.abs(tp.numpy() - pol).max() <= 1e-5 and np.abs(tv.numpy()[:, 0] - val).max() <= 1e-5
    # same module on the GPU through PyTorch-ROCm: also within tolerance
    with torch.no_grad():
        gp, gv = net.to(DEV)(x.to(DEV))
    assert np.abs(gp.cpu().numpy() - pol).max() <= 1e-5 and np.abs(gv.cpu().numpy()[:, 0] - val).max() <= 1e-5


def test_net_metric_shape_vs_oracle_and_torch():
    """The shape the metric is quoted on (config.py:187-193 defaults: F=256, R=10, V=256; agent/model.py:28-72), 64
    positions harvested from play: k_conv3x3_wide forward == the C oracle bit for bit on sampled positions, and every
    position within 1e-5 of the fp32 torch graph on CPU and through PyTorch-ROCm (20 stacked 256-channel convolutions
    is where f32 ordering differences accumulate)."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    net = ReversiNet(256, 10, 256).keras_init_(5).randomize_bn_(6)
    blob = net.to_blob()
    own, enemy = _harvested_positions(64, 3)
    pol, val = DeviceNet(blob, DEV).predict_bitboards(torch.from_numpy(own.view(np.int64)).to(DEV),
                                                      torch.from_numpy(enemy.view(np.int64)).to(DEV))
    pol, val = pol.cpu().numpy(), val.cpu().numpy()
    lib = O.load_ext()
    for i in (0, 31, 63):   # 0.76 GMAC each on one host core
        p = np.zeros(64, np.float32)
        v = np.zeros(1, np.float32)
        assert lib.orc_net_forward(blob, len(blob), int(own[i]), int(enemy[i]), p.ctypes.data, v.ctypes.data) == 0
        assert np.array_equal(p.view(np.uint32), pol[i].view(np.uint32)) and v.view(np.uint32)[0] == val[i:i + 1].view(np.uint32)[0], i
    bits = lambda a: ((a[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float32)
    x = torch.from_numpy(np.stack([bits(own), bits(enemy)], axis=1).reshape(-1, 2, 8, 8))
    with torch.no_grad():
        tp, tv = net(x)
        gp, gv = net.to(DEV)(x.to(DEV))
    ep, ev = np.abs(tp.numpy() - pol).max(), np.abs(tv.numpy()[:, 0] - val).max()
    gep, gev = np.abs(gp.cpu().numpy() - pol).max(), np.abs(gv.cpu().numpy()[:, 0] - val).max()
    print(f"F256 R10 V256, 64 positions: max |dp| {ep:.2e} |dv| {ev:.2e} vs torch CPU; {gep:.2e} {gev:.2e} vs torch ROCm")
    assert ep <= 1e-5 and ev <= 1e-5 and gep <= 1e-5 and gev <= 1e-5


@pytest.mark.parametrize("shape,n", [((256, 10, 256), 67), ((128, 2, 64), 9)])
def test_net_f16x3_within_tolerance_of_torch_and_exact_kernel(shape, n):
    """raznet-forward-v2 (csrc/raz_net_f16x3.hip: the 3x3 trunk on the f16 matrix cores, every operand split into two halfs,
    3 MFMAs per product, f32 accumulation) on the metric's shape, ragged batch with an active mask: within 1e-5 of the fp32
    torch graph (the north star's tolerance) AND of the exact-f32 kernel (raznet-forward-v1, itself bit-equal to the oracle);
    skipped positions stay untouched; no activation left the f16 range; results do not depend on the batch a position is in."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    net = ReversiNet(*shape).keras_init_(5).randomize_bn_(6)
    blob = net.to_blob()
    own, enemy = _harvested_positions(n, 11)
    o, e = torch.from_numpy(own.view(np.int64)).to(DEV), torch.from_numpy(enemy.view(np.int64)).to(DEV)
    act = torch.from_numpy((np.arange(n) % 7 != 3).astype(np.uint8)).to(DEV)
    v2 = DeviceNet(blob, DEV, kernel="f16x3")
    p2, q2 = v2.predict_bitboards(o, e, active=act)
    p1, q1 = DeviceNet(blob, DEV, kernel="f32").predict_bitboards(o, e, active=act)
    assert v2.range_ok()
    on = act.bool()
    assert bool((p2[~on] == 0).all()) and bool((q2[~on] == 0).all())
    d12p, d12v = float((p2 - p1)[on].abs().max()), float((q2 - q1)[on].abs().max())
    bits = lambda a: ((a[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float32)
    x = torch.from_numpy(np.stack([bits(own), bits(enemy)], axis=1).reshape(-1, 2, 8, 8))
    with torch.no_grad():
        tp, tv = net(x)
    dtp = float((p2.cpu() - tp)[on.cpu()].abs().max())
    dtv = float((q2.cpu() - tv[:, 0])[on.cpu()].abs().max())
    print(f"f16x3 {shape}: max |dp| {dtp:.2e} |dv| {dtv:.2e} vs fp32 torch; {d12p:.2e} {d12v:.2e} vs the exact-f32 kernel")
    assert dtp <= 1e-5 and dtv <= 1e-5 and d12p <= 1e-5 and d12v <= 1e-5
    # batch invariance: a position alone == the same position inside the batch, bit for bit
    for i in (0, n - 1):
        pa, qa = v2.predict_bitboards(o[i:i + 1], e[i:i + 1])
        if bool(on[i]):
            assert torch.equal(pa[0].view(torch.int32), p2[i].view(torch.int32)) and torch.equal(qa.view(torch.int32), q2[i:i + 1].view(torch.int32))


def test_net_f16x3_rows_that_leave_the_f16_range_are_evaluated_by_the_exact_f32_chains():
    """raznet-forward-v2's range repair on the device (csrc/raz_net.hip k_net_wave_repair; the emulated form of this test is in
    tests/test_net_emu.py): a 256-filter net whose stem activations are 10^4 x the discs around a square.  In a batch of 4096
    harvested positions with 20 completely filled boards mixed in, the filled rows (beyond 60000) equal the exact-f32 kernels' outputs bit
    for bit, every other row equals the v2 forward of the in-range rows alone (bitwise batch invariance), the sticky flag stays down
    (20 <= 32 rows in one forward) and the repair counter says 20.  Then a forward in which most rows are out of range raises the flag:
    that is the case the worker answers by moving the block to the f32 kernels (tests/test_worker_scale_gpu.py)."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_sweep import harvest_positions
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    dev = torch.device(DEV)
    net = ReversiNet(256, 2, 64).keras_init_(6)
    with torch.no_grad():
        net.stem.conv.weight.fill_(1.0e4)
        net.stem.conv.bias.zero_()
        for blk in net.res:
            for cb in blk:
                cb.conv.weight.mul_(1.0e-7)
    blob = net.to_blob()
    n, n_full = 4096, 20
    black, white, player, _ = harvest_positions(n, 41, dev, None)
    own = torch.where(player == 1, black, white)
    enemy = torch.where(player == 1, white, black)
    # early positions only (at most 3 discs around any square would be too strict for real play: keep the rows whose stem stays in range
    # by construction instead - boards of <= 5 discs per 3x3 neighbourhood, i.e. < 60000)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    occ = (((own | enemy)[:, None] >> sh) & 1).float().reshape(-1, 1, 8, 8)
    crowd = torch.nn.functional.conv2d(occ, torch.ones(1, 1, 3, 3, device=dev), padding=1).amax(dim=(1, 2, 3))
    keep = crowd <= 5
    own, enemy = own[keep].contiguous(), enemy[keep].contiguous()
    assert own.numel() >= 100
    g = torch.Generator(device="cpu").manual_seed(3)
    full_own = torch.randint(-2**63, 2**63 - 1, (n_full,), generator=g, dtype=torch.int64).to(dev)
    where = torch.randperm(own.numel() + n_full, generator=g).to(dev)
    mixed_own = torch.cat([own, full_own])[where].contiguous()
    mixed_enemy = torch.cat([enemy, ~full_own])[where].contiguous()
    is_full = (where >= own.numel())
    v2, v1 = DeviceNet(blob, dev, kernel="f16x3"), DeviceNet(blob, dev, kernel="f32")
    p2, q2 = v2.predict_bitboards(mixed_own, mixed_enemy)
    p1, q1 = v1.predict_bitboards(mixed_own, mixed_enemy)
    assert torch.isfinite(p2).all() and torch.isfinite(q2).all()
    assert torch.equal(p2[is_full].view(torch.int32), p1[is_full].view(torch.int32)) and torch.equal(q2[is_full].view(torch.int32), q1[is_full].view(torch.int32))
    assert v2.range_stats() == (True, n_full)
    alone = DeviceNet(blob, dev, kernel="f16x3")
    pa, qa = alone.predict_bitboards(own, enemy)
    assert alone.range_stats() == (True, 0)
    idx = where[~is_full]   # mixed row j holds own[where[j]]
    assert torch.equal(p2[~is_full].view(torch.int32), pa[idx].view(torch.int32)) and torch.equal(q2[~is_full].view(torch.int32), qa[idx].view(torch.int32))
    # most rows out of range in ONE forward: answers are still the exact-f32 ones, and the sticky flag tells the caller to change kernels
    many = torch.cat([full_own, full_own ^ 0x5555, full_own ^ 0x3333]).contiguous()
    pm, qm = v2.predict_bitboards(many, ~many)
    p1m, q1m = v1.predict_bitboards(many, ~many)
    assert torch.equal(pm.view(torch.int32), p1m.view(torch.int32)) and torch.equal(qm.view(torch.int32), q1m.view(torch.int32))
    assert v2.range_stats() == (False, n_full + 3 * n_full) and not v2.range_ok()


def test_net_f16x3_accuracy_on_4096_positions_of_three_nets():
    """raznet-forward-v2 on the metric's shape (256x10), 4096 positions from random play (tools/bench_sweep.harvest_positions) x
    three weight / BatchNorm-statistics variants of tools/check_net_accuracy.py: (1) the bench net (Keras initialisers, seed 0), (2)
    BN statistics in [0.5, 1.5], (3) BN gamma / variance spread over 10^+-0.5 - the spread a trained checkpoint can show, where two
    fp32 evaluations of the graph differ by more than 1e-5 from each other.  For (1) and (2): within the north star's 1e-5 of fp32
    torch (ROCm).  For all three, against the f64 evaluation of the same graph on 1024 of the positions: mean and maximum error no
    larger than 2.5 x fp32 torch's own (measured here: mean 2.9e-8 / 8.9e-7 against torch's 1.4e-8 / 5.5e-7, maximum 1.4e-7 / 5.1e-6
    against 9.0e-8 / 2.8e-6 for the first two nets; round 3's 131 072-position sweep had maxima of 2.1e-8 / 5.6e-6 / 1.25e-4 against
    2.0e-8 / 6.4e-6 / 1.26e-4) - i.e. the split operands and the matrix core's summation order cost about one more fp32 evaluation's
    worth of rounding, well inside the 1e-5 budget.  The range flag stays clear."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_sweep import harvest_positions
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    dev = torch.device(DEV)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    n, n64 = 4096, 1024
    black, white, player, _ = harvest_positions(n, 99, dev)
    own = torch.where(player == 1, black, white)
    enemy = torch.where(player == 1, white, black)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    planes = torch.stack([((own[:, None] >> sh) & 1), ((enemy[:, None] >> sh) & 1)], dim=1).float().reshape(-1, 2, 8, 8)
    for name, seed, bn_seed, decades, within_1e5 in (("bench net", 0, None, 0.0, True), ("BN stats in [0.5, 1.5]", 5, 6, 0.0, True),
                                                      ("BN gamma / var over 10^+-0.5", 11, 12, 0.5, False)):
        net = ReversiNet(256, 10, 256).keras_init_(seed)
        if bn_seed is not None:
            net.randomize_bn_(bn_seed, decades=decades)
        net.eval()
        n32 = ReversiNet(256, 10, 256)
        n32.load_state_dict(net.state_dict())
        n32 = n32.to(dev).eval()
        n64net = ReversiNet(256, 10, 256)
        n64net.load_state_dict(net.state_dict())
        n64net = n64net.double().to(dev).eval()
        with torch.no_grad():
            tp, tv = n32(planes)
            dp, dv = n64net(planes[:n64].double())
        dn = DeviceNet(net.to_blob(), dev, kernel="f16x3")
        p, v = dn.predict_bitboards(own, enemy)
        assert dn.range_ok(), name
        e32 = max(float((p - tp).abs().max()), float((v - tv[:, 0]).abs().max()))
        e64 = max(float((p[:n64].double() - dp).abs().max()), float((v[:n64].double() - dv[:, 0]).abs().max()))
        t64 = max(float((tp[:n64].double() - dp).abs().max()), float((tv[:n64, 0].double() - dv[:, 0]).abs().max()))
        m64 = float((p[:n64].double() - dp).abs().mean()) + float((v[:n64].double() - dv[:, 0]).abs().mean())
        mt64 = float((tp[:n64].double() - dp).abs().mean()) + float((tv[:n64, 0].double() - dv[:, 0]).abs().mean())
        print(f"f16x3, 4096 positions, {name}: {e32:.2e} vs fp32 torch; vs f64 max {e64:.2e} mean {m64:.2e} (fp32 torch itself: max {t64:.2e} mean {mt64:.2e})")
        if within_1e5:
            assert e32 <= 1e-5, (name, e32)
        assert m64 <= 2.5 * mt64 + 2e-8, (name, m64, mt64)
        assert e64 <= 2.5 * t64 + 1e-7, (name, e64, t64)
        del n32, n64net, dn
        torch.cuda.empty_cache()


def test_net_f16x3_range_flag():
    """A net whose activations leave the f16 range must never return garbage silently: up to 32 such rows per forward are evaluated
    by the exact-f32 chains inside the forward (answers == the f32 kernels', bit for bit; raz_net_range_stats counts them), a forward
    with more raises the sticky flag (raz_net_range_check) while still answering every row exactly."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    net = ReversiNet(128, 1, 64).keras_init_(5)
    with torch.no_grad():
        net.stem.conv.weight.mul_(1.0e6)
    blob = net.to_blob()
    v1 = DeviceNet(blob, DEV, kernel="f32")
    for n, in_range in ((8, True), (40, False)):
        own, enemy = _harvested_positions(n, 2)
        o, e = torch.from_numpy(own.view(np.int64)).to(DEV), torch.from_numpy(enemy.view(np.int64)).to(DEV)
        v2 = DeviceNet(blob, DEV, kernel="f16x3")
        assert v2.range_ok()
        p2, q2 = v2.predict_bitboards(o, e)
        p1, q1 = v1.predict_bitboards(o, e)
        assert torch.equal(p2.view(torch.int32), p1.view(torch.int32)) and torch.equal(q2.view(torch.int32), q1.view(torch.int32))
        assert v2.range_stats() == (in_range, n) and v2.range_ok() == in_range


@pytest.mark.parametrize("par", [1, 4])
def test_engine_on_f16x3_net_equals_oracle_given_the_nets_outputs(par):
    """Games on raznet-forward-v2 (par = parallel_search_num: k_tree / the slot kernel k_tree_par): the engine (wide net, split-f16 trunk, cross-game leaf batches) == the CPU oracle's
    games when the oracle evaluates leaves through the SAME device net via the reference's NN seam
    (ReversiPlayer(api=...), agent/player.py:41,346) - every action, root N and W, bit for bit."""
    import types
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    blob = ReversiNet(128, 1, 64).keras_init_(7).randomize_bn_(8).to_blob()
    play = types.SimpleNamespace(
        simulation_num_per_move=14, share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=400,
        start_rethinking_turn=8, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=4, virtual_loss=3,
        parallel_search_num=par, resign_threshold=-0.9, allowed_resign_turn=50, disable_resignation_rate=0.1,
        use_solver_turn=0, use_solver_turn_in_simulation=0)
    cfg = types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
    dnet = DeviceNet(blob, DEV, kernel="f16x3")
    n = 12
    eng = SelfPlayEngine(cfg, dnet, n_games=n, seed=3, sims_hint=14, record_root_w=True)
    eng.start(first_game_id=40, sims_per_move=14)
    eng.run(chunk=64)
    recs = eng.records()

    def nn(own, enemy):
        to = lambda v: torch.tensor([v - (1 << 64) if v >= 1 << 63 else v], dtype=torch.int64, device=DEV)
        p, v = dnet.predict_bitboards(to(own), to(enemy))
        return p[0].cpu().numpy(), float(v[0].item())
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=par)
    for i in (0, 5, 11):
        plies, summ = O.selfplay_game(ocfg, None, 3, 40 + i, 14, nn=nn)
        _compare_game(f"f16x3/par{par}/{40 + i}", recs[i][0], recs[i][1], plies, summ["winner"])
    assert dnet.range_ok()


def _harvested_positions(n, seed):
    """Positions the way SURVEY 8(d) asks for them: random playouts from the start, stopped at a random ply
    (oracle env = test infrastructure); returned from the mover's view."""
    rng = np.random.default_rng(seed)
    own, enemy = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    orc = O.load()
    for i in range(n):
        env = O.OrcEnv()
        orc.orc_env_reset(env)
        for _ in range(int(rng.integers(0, 58))):
            if env.done:
                break
            o, e = (env.black, env.white) if env.next_player == 1 else (env.white, env.black)
            legal = orc.orc_find_correct_moves(o, e)
            moves = [s for s in range(64) if legal >> s & 1]
            prev = (env.black, env.white, env.next_player)
            orc.orc_env_step(env, int(moves[rng.integers(0, len(moves))]))
            if env.done:   # keep the last live position
                orc.orc_env_update(env, prev[0], prev[1], prev[2])
                break
        own[i], enemy[i] = (env.black, env.white) if env.next_player == 1 else (env.white, env.black)
    return own, enemy


def _compare_game(tag, eng_plies, eng_sum, ref_plies, ref_winner, check_w=True):
    assert [p["action"] for p in eng_plies] == [p["action"] for p in ref_plies], tag
    assert eng_sum["winner"] == ref_winner, tag
    for i, (a, b) in enumerate(zip(eng_plies, ref_plies)):
        assert a["player"] == b["player"] and a["own"] == b["own"] and a["enemy"] == b["enemy"], (tag, i)
        assert a["root_n"] == b["root_n"], (tag, i)
        if check_w:
            assert a["root_w"] == b["root_w"], (tag, i)
        assert a["has_row"] == b["has_row"], (tag, i)
        assert a["solved"] == b.get("solved", False), (tag, i)
        if a["action"] >= 0:
            assert a["n"] == b["n"] and a["q"] == b["q"], (tag, i)
        if b["has_row"]:
            assert a["saved_policy"] == b["saved_policy"], (tag, i)


def test_engine_reproduces_reference_golden_games(golden, blob):
    """Every golden game (7 config variants incl. shared tree, re-thinking, resignation) is replayed
    by the engine alone in its slot and must match the unmodified reference's record exactly."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    dnet = DeviceNet(blob, DEV)
    for g in golden["games"]:
        cfg = config_of(g)
        eng = SelfPlayEngine(cfg, dnet, n_games=1, seed=g["seed"], sims_hint=g["sims_per_move"], record_root_w=True)
        eng.start(first_game_id=g["game_id"], sims_per_move=g["sims_per_move"])
        eng.run(chunk=256)
        (plies, summ), = eng.records(save_policy_of_tau_1=g["resolved_play_data"]["save_policy_of_tau_1"])
        ref = [dict(p, own=int(p["own"], 16), enemy=int(p["enemy"], 16), root_n=dense(p["root_n"]),
                    root_w=dense(p["root_w"]), saved_policy=dense(p["saved_policy"]) if p["has_row"] else None)
               for p in g["plies"]]
        _compare_game(f'{g["variant"]}/{g["game_id"]}', plies, summ, ref, g["winner"])
        assert (bool(summ["resigned_black"]), bool(summ["resigned_white"])) == (g["resigned_black"], g["resigned_white"])
        assert (summ["black"], summ["white"]) == (int(g["black"], 16), int(g["white"], 16))


def test_engine_reproduces_reference_games_at_other_dirichlet_alphas(blob):
    """B6 with alphas other than 0.5 (the lane-parallel rejection sampler of raz_engine.hip select_action, not the
    Box-Muller pairs): the reference's games at dirichlet_alpha 0.3, 1.0 and 0.03 replayed bit-exactly on the device."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    alpha = load_mcts_golden("mcts_alpha_games.json")
    dnet = DeviceNet(blob, DEV)
    seen = set()
    for g in alpha["games"]:
        cfg = config_of(g)
        seen.add(cfg.play.dirichlet_alpha)
        eng = SelfPlayEngine(cfg, dnet, n_games=1, seed=g["seed"], sims_hint=g["sims_per_move"], record_root_w=True)
        eng.start(first_game_id=g["game_id"], sims_per_move=g["sims_per_move"])
        eng.run(chunk=256)
        (plies, summ), = eng.records(save_policy_of_tau_1=g["resolved_play_data"]["save_policy_of_tau_1"])
        ref = [dict(p, own=int(p["own"], 16), enemy=int(p["enemy"], 16), root_n=dense(p["root_n"]),
                    root_w=dense(p["root_w"]), saved_policy=dense(p["saved_policy"]) if p["has_row"] else None)
               for p in g["plies"]]
        _compare_game(f'{g["variant"]}/{g["game_id"]}', plies, summ, ref, g["winner"])
        assert (summ["black"], summ["white"]) == (int(g["black"], 16), int(g["white"], 16))
    assert seen == {0.3, 1.0, 0.03}


@pytest.mark.parametrize("alpha,eps", [(0.3, 0.25), (1.0, 0.25), (0.03, 0.5), (0.75, 0.1)])
def test_engine_batch_vs_oracle_at_other_dirichlet_alphas(golden, blob, alpha, eps):
    """40 concurrent games per alpha (shared tree, ch5-like settings) == the oracle's games, bit for bit."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in golden["games"] if g["variant"] == "mini_shared")
    cfg = config_of(g0)
    cfg.play.dirichlet_alpha, cfg.play.noise_eps = alpha, eps
    n, sims = 40, 22
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=n, seed=61, sims_hint=sims, record_root_w=True)
    eng.start(first_game_id=3000, sims_per_move=sims)
    eng.run(chunk=128)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg)
    for i in range(0, n, 2):
        plies, summ = O.selfplay_game(ocfg, blob, 61, 3000 + i, sims)
        _compare_game(f"alpha{alpha}/{3000 + i}", recs[i][0], recs[i][1], plies, summ["winner"])


@pytest.mark.parametrize("variant", ["agz", "mini_shared"])
def test_engine_batch_vs_oracle_many_games(golden, blob, variant):
    """64 concurrent games (ids 1000..1063, mixed sims per move) == 64 independent oracle games:
    results do not depend on batching or slot order."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in golden["games"] if g["variant"] == variant)
    cfg = config_of(g0)
    n = 64
    sims = np.array([12 + (i % 5) * 7 for i in range(n)], dtype=np.uint32)
    dnet = DeviceNet(blob, DEV)
    eng = SelfPlayEngine(cfg, dnet, n_games=n, seed=77, sims_hint=int(sims.max()), record_root_w=True)
    eng.start(first_game_id=1000, sims_per_move=sims)
    st = eng.run(chunk=128)
    recs = eng.records(save_policy_of_tau_1=g0["resolved_play_data"]["save_policy_of_tau_1"])
    ocfg = O.play_cfg_from_config(cfg)
    total = 0
    for i in range(0, n, 3):
        plies, summ = O.selfplay_game(ocfg, blob, 77, 1000 + i, int(sims[i]))
        _compare_game(f"{variant}/{1000 + i}", recs[i][0], recs[i][1], plies, summ["winner"])
        assert sum(p["sims"] for p in recs[i][0]) == summ["n_sims"]
    assert st["total_sims"] == sum(sum(p["sims"] for p in r[0]) for r in recs)


def test_engine_mirror_off_equals_mirror_on_unshared(golden, blob):
    """share=False: skipping the colour-mirrored writes changes nothing (they are dead)."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in golden["games"] if g["variant"] == "agz")
    cfg = config_of(g0)
    dnet = DeviceNet(blob, DEV)
    out = []
    for mirror in (False, True):
        eng = SelfPlayEngine(cfg, dnet, n_games=8, seed=5, sims_hint=20, mirror_updates=mirror, record_root_w=True)
        eng.start(first_game_id=0, sims_per_move=20)
        eng.run(chunk=128)
        out.append(eng.records(save_policy_of_tau_1=False))
    for (pa, sa), (pb, sb) in zip(*out):
        assert pa == pb and sa == sb


@pytest.mark.parametrize("variant,pool", [("mini_shared", 1024), ("agz", 512)])
def test_engine_with_node_pruning_equals_oracle(golden, blob, variant, pool):
    """A node pool far too small for whole games (forces several k_gc prunes per game) must not
    change anything: 48 games == the oracle, bit for bit."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in golden["games"] if g["variant"] == variant)
    cfg = config_of(g0)
    n, sims = 48, 24
    dnet = DeviceNet(blob, DEV)
    eng = SelfPlayEngine(cfg, dnet, n_games=n, seed=31, nodes_per_game=pool, record_root_w=True)
    eng.start(first_game_id=500, sims_per_move=sims)
    eng.run(chunk=32)
    assert eng.gc_runs >= 2
    recs = eng.records(save_policy_of_tau_1=g0["resolved_play_data"]["save_policy_of_tau_1"])
    ocfg = O.play_cfg_from_config(cfg)
    for i in range(0, n, 2):
        plies, summ = O.selfplay_game(ocfg, blob, 31, 500 + i, sims)
        _compare_game(f"gc/{variant}/{500 + i}", recs[i][0], recs[i][1], plies, summ["winner"])


@pytest.mark.parametrize("par,fused", [(1, False), (4, False), (1, True), (3, True)])
def test_engine_solves_suspended_on_the_smallest_budget_equal_the_oracle(golden, blob, par, fused):
    """raz_engine_config.reserved bits 16-23 = 1: a game may spend 64 solver iterations per launch (default 384), so every end-game
    solve of more than a few hundred nodes - the exact ones at the root and the win/loss ones in the middle of a descent, whose
    simulation then waits in its slot - is suspended and resumed many times, on k_tree (par 1), k_tree_par (4), k_tree_net and
    k_tree_par_net (fused).  Records == the oracle's, which has no budget (mini.yml as shipped, resignation off, 12 games)."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in golden["games"] if g["variant"] == "mini_solver_noresign")
    cfg = config_of(g0)
    cfg.play.parallel_search_num = par
    n, sims = 12, 14
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=n, seed=43, sims_hint=sims, record_root_w=True, fused=fused, solver_budget=64)
    assert (int(eng.cfg.reserved) >> 16) & 0xff == 1
    eng.start(first_game_id=700, sims_per_move=sims)
    eng.run(chunk=48)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=par) if par > 1 else O.play_cfg_from_config(cfg)
    nsolved = 0
    for i in range(0, n, 3):
        plies, summ = O.selfplay_game(ocfg, blob, 43, 700 + i, sims)
        _compare_game(f"budget64/par{par}/{700 + i}", recs[i][0], recs[i][1], plies, summ["winner"])
        nsolved += sum(p["solved"] for p in plies)
    assert nsolved > 0


@pytest.mark.parametrize("par,waves,budget", [(2, 8, 64), (3, 24, 64), (4, 2048, 128)])
def test_engine_solver_pool_of_any_size_equals_the_oracle(golden, blob, par, waves, budget):
    """The end-game solver's pool (csrc/raz_solver_pool.h) on the device: 16 games post their solves into ONE pool of `waves` worker
    waves - 8 (each slice's single wave serves every game: every search is parked again and again), 24, 2048 (more lanes than tasks:
    the launch runs in two rounds of resident waves) - at parallel_search_num 2, 3 and 4 (2 and 3 with the solver on were not covered
    before round 5).  Records == the oracle's; the pool's counters say that it did the solving."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in golden["games"] if g["variant"] == "mini_solver_noresign")
    cfg = config_of(g0)
    cfg.play.parallel_search_num = par
    n, sims = 16, 12
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=n, seed=43, sims_hint=sims, record_root_w=True, solver_budget=budget, solver_pool_waves=waves)
    eng.start(first_game_id=900, sims_per_move=sims)
    eng.run(chunk=48)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=par)
    nsolved = 0
    for i in range(0, n, 4):
        plies, summ = O.selfplay_game(ocfg, blob, 43, 900 + i, sims)
        _compare_game(f"pool{waves}/par{par}/{900 + i}", recs[i][0], recs[i][1], plies, summ["winner"])
        nsolved += sum(p["solved"] for p in plies)
    st = eng.solver_stats()
    assert nsolved > 0 and st["answers"] > 0 and st["subtrees_finished"] > 0 and st["requests_posted"] >= st["answers"]


@pytest.mark.parametrize("par", [1, 4])
def test_engine_solves_of_11_to_13_empties_equal_the_oracle(golden, blob, par):
    """The lane-parallel solver on LARGE task trees (three plies below positions of 11-13 empties: up to ~150 level-2 nodes and
    ~1500 tasks; the goldens' solves have at most 10): (a) non-exact - positions taken up with the in-simulation solver from turn 46
    and no root solver, three simulations (the first solves the root position itself, the next ones its children); (b) exact -
    positions of 11 empties decided by the root solver (use_solver_turn 46).  Decided move, root statistics and (b) the solved flag
    == the oracle's for the same id from the same position.  (Emulator form of (a): tests/test_engine_emu.py.)"""
    import random
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    from reversi_alpha_zero_amd.lib import bitboard as bb
    rng = random.Random(2024 + par)

    def position(empties):
        while True:
            b, w, p = 0x0000000810000000, 0x0000001008000000, 1
            want = 64 - empties
            while bin(b | w).count("1") < want:
                own, enemy = (b, w) if p == 1 else (w, b)
                legal = int(bb.find_correct_moves(own, enemy))
                if not legal:
                    if not int(bb.find_correct_moves(enemy, own)):
                        break
                    p = 3 - p
                    continue
                a = rng.choice([i for i in range(64) if legal >> i & 1])
                fl = int(bb.calc_flip(a, own, enemy))
                own, enemy = (own ^ fl) | (1 << a), enemy ^ fl
                b, w = (own, enemy) if p == 1 else (enemy, own)
                p = 3 - p
            own, enemy = (b, w) if p == 1 else (w, b)
            if bin(b | w).count("1") == want and int(bb.find_correct_moves(own, enemy)):
                return b, w, p
    for mode, root_turn, insim_turn, sims, empties in (("non-exact", 0, 46, 3, (11, 12, 13, 11, 12, 13, 12, 13)), ("exact", 46, 46, 2, (11, 11, 11, 11))):
        cfg = config_of(next(g for g in golden["games"] if g["variant"] == "mini_solver_noresign"))
        cfg.play.use_solver_turn, cfg.play.use_solver_turn_in_simulation = root_turn, insim_turn
        cfg.play.parallel_search_num, cfg.play.thinking_loop = par, 1
        starts = [position(e) for e in empties]
        eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=len(starts), seed=9, sims_hint=sims, record_root_w=True)
        eng.start(first_game_id=300, sims_per_move=sims)
        for g, (b, w, p) in enumerate(starts):
            eng.set_position(g, b, w, p, sims, enable_resign=False, one_move=True)
        for _ in range(100000):
            eng.step(16)
            if eng.stats()["idle_or_done"] >= len(starts):
                break
        recs = eng.records(save_policy_of_tau_1=True)
        ocfg = O.play_cfg_from_config(cfg, parallel_search_num=par) if par > 1 else O.play_cfg_from_config(cfg)
        for g, start in enumerate(starts):
            oplies, _ = O.selfplay_game(ocfg, blob, 9, 300 + g, sims, stop_after_plies=1, start=start)
            a, ref = recs[g][0][0], oplies[0]
            for key in ("player", "own", "enemy", "action", "root_n", "root_w", "n", "q"):
                assert a[key] == ref[key], (mode, g, key, a[key], ref[key])
            assert a["solved"] == ref.get("solved", False) == (mode == "exact"), (mode, g)


def test_engine_with_solver_batch_vs_oracle(golden, blob):
    """End-game solver on (mini.yml as shipped: exact at the root from turn 50, win/loss inside
    simulations from turn 50), resignation off so every game reaches the solver: 24 games == oracle."""
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    g0 = next(g for g in golden["games"] if g["variant"] == "mini_solver_noresign")
    cfg = config_of(g0)
    n, sims = 24, 14
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=n, seed=41, sims_hint=sims, record_root_w=True)
    eng.start(first_game_id=900, sims_per_move=sims)
    eng.run(chunk=64)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg)
    nsolved = 0
    for i in range(0, n, 2):
        plies, summ = O.selfplay_game(ocfg, blob, 41, 900 + i, sims)
        _compare_game(f"solver/{900 + i}", recs[i][0], recs[i][1], plies, summ["winner"])
        nsolved += sum(p["solved"] for p in plies)
    assert nsolved > 0


def test_model_api_predict_contract():
    """ReversiModelAPI.predict (agent/api.py:30-45 contract) on planes == the bitboard entry point."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.agent.api import ReversiModelAPI
    from reversi_alpha_zero_amd.engine import DeviceNet
    net = ReversiNet(16, 1, 16).keras_init_(4)
    api = ReversiModelAPI(None, net, DEV)
    own, enemy = _positions(9, 12)
    bits = lambda a: ((a[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8)
    x = np.stack([bits(own), bits(enemy)], axis=1).reshape(9, 2, 8, 8)
    p, v = api.predict(x)
    assert p.shape == (9, 64) and v.shape == (9, 1) and p.dtype == np.float32
    p1, v1 = api.predict(x[3])
    assert p1.shape == (64,) and v1.shape == (1,) and np.array_equal(p1, p[3]) and v1[0] == v[3, 0]
    pb, vb = DeviceNet(net.to_blob(), DEV).predict_bitboards(torch.from_numpy(own.view(np.int64)).to(DEV),
                                                             torch.from_numpy(enemy.view(np.int64)).to(DEV))
    assert np.array_equal(pb.cpu().numpy(), p) and np.array_equal(vb.cpu().numpy(), v[:, 0])
    with pytest.raises(AssertionError):
        api.predict(np.zeros((2, 8)))


@pytest.mark.parametrize("raw_path", [False, True])
def test_worker_files_equal_oracle_rows(golden, blob, tmp_path, raw_path):
    """(raw_path: the worker's production path - engine record arrays, JSON text written natively by
    raz_emit_game_rows_json - instead of per-ply Python objects and json.dump; same files.)
    The `self` worker end to end on the GPU: play 6 games (mini.yml settings as shipped, solver on),
    write play_*.json / GGF / game-idx the way worker/self_play.py does, and check the file content is
    exactly what the reference's row construction yields for the oracle's games, and that it loads the
    way the reference's trainer reads it (worker/optimize.py:214-231)."""
    import json
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.worker.self_play import BatchedSelfPlayWorker
    from reversi_alpha_zero_amd.lib.data_helper import get_game_data_filenames, read_game_data_from_file
    from reversi_alpha_zero_amd.lib.bitboard import bit_to_array
    from oracle_util import rows_of_game
    g0 = next(g for g in golden["games"] if g["variant"] == "config0_mini_yml_100sims")
    cfg = Config()
    cfg.play.update(g0["resolved_play"])
    cfg.play.schedule_of_simulation_num_per_move = [(0, 12)]
    cfg.play_data.update(dict(g0["resolved_play_data"], nb_game_in_file=2, nb_game_in_ggf_file=3))
    rc = cfg.resource
    rc.data_dir = str(tmp_path)
    rc.play_data_dir = str(tmp_path / "play_data")
    rc.self_play_ggf_data_dir = str(tmp_path / "ggf")
    rc.model_dir = str(tmp_path / "model")
    rc.next_generation_model_dir = str(tmp_path / "model" / "next")
    rc.log_dir = str(tmp_path / "logs")
    rc.project_dir = str(tmp_path)
    rc.force_simulation_num_file = str(tmp_path / ".force-sim")
    rc.self_play_game_idx_file = str(tmp_path / ".self-play-game-idx")
    rc.create_directories()
    w = BatchedSelfPlayWorker(cfg, blob, games_in_flight=6, seed=3, device=DEV)
    if raw_path:
        paths = w.emit_raw(w.play_batch_raw(first_game_idx=0), first_local_idx=1)
    else:
        paths = w.emit(w.play_batch(first_game_idx=0), first_local_idx=1)
    assert len(paths) == 3                                   # nb_game_in_file = 2
    files = get_game_data_filenames(rc)
    got = [row for f in files for row in read_game_data_from_file(f)]
    ocfg = O.play_cfg_from_config(cfg)
    exp = []
    for i in range(6):
        plies, summ = O.selfplay_game(ocfg, blob, 3, i, 12)
        exp += rows_of_game(plies, summ["winner"])
    assert json.dumps(got) == json.dumps(exp)
    # the trainer's view of the data (worker/optimize.py:214-231)
    state = np.array([[bit_to_array(r[0][0], 64).reshape(8, 8), bit_to_array(r[0][1], 64).reshape(8, 8)] for r in got])
    policy = np.array([r[1] for r in got])
    z = np.array([r[2] for r in got])
    assert state.shape == (len(got), 2, 8, 8) and policy.shape == (len(got), 64) and z.shape == (len(got),)
    assert np.allclose(policy.sum(axis=1), 1.0) and set(np.unique(z)) <= {-1, 0, 1}
    assert len(list((tmp_path / "ggf").iterdir())) >= 1


def test_engine_wide_net_vs_oracle():
    """ch5-shaped path (wide-net implicit-GEMM kernels, per-slice activation scratch, 3 streams):
    256 games with a 128-filter net; two of them replayed by the oracle, bit for bit."""
    import types
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
    blob = ReversiNet(128, 1, 32).keras_init_(21).randomize_bn_(22).to_blob()
    play = types.SimpleNamespace(
        share_mtcs_info_in_self_play=True, thinking_loop=1, required_visit_to_decide_action=400,
        start_rethinking_turn=8, c_puct=5, noise_eps=0.25, dirichlet_alpha=0.5, change_tau_turn=4,
        virtual_loss=3, parallel_search_num=1, resign_threshold=-0.9, allowed_resign_turn=50,
        disable_resignation_rate=0.1, use_solver_turn=0, use_solver_turn_in_simulation=0)   # ch5.yml, solver off
    cfg = types.SimpleNamespace(play=play, play_data=types.SimpleNamespace(save_policy_of_tau_1=True))
    n, sims = 256, 8
    eng = SelfPlayEngine(cfg, DeviceNet(blob, DEV), n_games=n, seed=9, sims_hint=sims, record_root_w=True)
    eng.start(first_game_id=0, sims_per_move=sims)
    eng.run(chunk=128)
    recs = eng.records(save_policy_of_tau_1=True)
    ocfg = O.play_cfg_from_config(cfg)
    for i in (0, 255):
        plies, summ = O.selfplay_game(ocfg, blob, 9, i, sims)
        _compare_game(f"wide/{i}", recs[i][0], recs[i][1], plies, summ["winner"])


def _facade_game(cfg, model, seed, gid, sims, callback=None):
    """SelfPlayWorker.start_game (worker/self_play.py:139-175) driven through the ReversiPlayer drop-in."""
    from reversi_alpha_zero_amd.agent.player import ReversiPlayer
    from reversi_alpha_zero_amd.env.reversi_env import ReversiEnv, Player, Winner
    from reversi_alpha_zero_amd._rng import rng_pair
    cfg.play.simulation_num_per_move = sims
    enable_resign = cfg.play.disable_resignation_rate <= rng_pair(seed, gid, 3, 0)[0]
    info = ReversiPlayer.create_mtcs_info(seed=seed, game_id=gid, device=DEV)
    black = ReversiPlayer(cfg, model, enable_resign=enable_resign, mtcs_info=info)
    white = ReversiPlayer(cfg, model, enable_resign=enable_resign, mtcs_info=info)
    env = ReversiEnv().reset()
    evals = []
    while not env.done:
        if env.next_player == Player.black:
            ae = black.action_with_evaluation(env.board.black, env.board.white, callback_in_mtcs=callback)
        else:
            ae = white.action_with_evaluation(env.board.white, env.board.black, callback_in_mtcs=callback)
        evals.append((env.next_player.value, ae))
        env.step(ae.action)
    black_win = {Winner.black: 1, Winner.white: -1}.get(env.winner, 0)
    black.finish_game(black_win)
    white.finish_game(-black_win)
    return env, black, white, evals


@pytest.mark.parametrize("variant", ["config0_mini_yml_100sims", "agz"])
def test_reversi_player_facade_game_equals_oracle(golden, blob, variant):
    """A whole game played through the ReversiPlayer drop-in (agent/player.py API: two players, shared
    MCTSInfo handle, ReversiEnv, finish_game) is the oracle's game of the same (seed, game id): same
    actions, n, q, resign flags and the same training rows black.moves + white.moves."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from oracle_util import rows_of_game
    g0 = next(g for g in golden["games"] if g["variant"] == variant)
    cfg = Config()
    cfg.play.update(g0["resolved_play"])
    cfg.play_data.update(g0["resolved_play_data"])
    meta = golden["net"]
    net = ReversiNet(meta["filters"], meta["res_layers"], meta["value_fc"]).keras_init_(meta["keras_init_seed"])
    net.randomize_bn_(meta["randomize_bn_seed"])
    seed, gid, sims = 5, 40, 20
    env, black, white, evals = _facade_game(cfg, net, seed, gid, sims)
    plies, summ = O.selfplay_game(O.play_cfg_from_config(cfg), blob, seed, gid, sims)
    assert len(evals) == len(plies)
    for (pl, ae), p in zip(evals, plies):
        assert pl == p["player"]
        assert (ae.action if ae.action is not None else -1) == p["action"]
        if ae.action is not None:
            assert float(ae.n) == p["n"] and float(ae.q) == p["q"]
    assert {"black": 1, "white": 2, "draw": 3}[env.winner.name] == summ["winner"]
    assert (black.resigned, white.resigned) == (bool(summ["resigned_black"]), bool(summ["resigned_white"]))
    import json
    assert json.dumps(black.moves + white.moves) == json.dumps(rows_of_game(plies, summ["winner"]))
    # MCTSInfo introspection: the root statistics of the last searched ply are readable by key
    last = next(p for p in reversed(plies) if p["has_row"])
    who = black if last["player"] == 1 else white
    hist = who.ask_thought_about(last["own"], last["enemy"])
    assert hist is not None and hist.action == last["action"] and len(hist.values) == 64


def test_reversi_player_callback_and_stop_thinking(golden, blob):
    """callback_in_mtcs is invoked with (q list, n list) during the search, and stop_thinking() from
    the callback ends the search early (agent/player.py:163-164,196-199,224-227)."""
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.agent.player import ReversiPlayer, CallbackInMCTS
    g0 = next(g for g in golden["games"] if g["variant"] == "agz")
    cfg = Config()
    cfg.play.update(g0["resolved_play"])
    cfg.play_data.update(g0["resolved_play_data"])
    cfg.play.simulation_num_per_move = 400
    meta = golden["net"]
    net = ReversiNet(meta["filters"], meta["res_layers"], meta["value_fc"]).keras_init_(meta["keras_init_seed"])
    p = ReversiPlayer(cfg, net, enable_resign=False)
    calls = []

    def cb(q, n):
        calls.append(sum(n))
        if len(calls) == 3:
            p.stop_thinking()

    from reversi_alpha_zero_amd.lib.bitboard import find_correct_moves
    from reversi_alpha_zero_amd.env.reversi_env import ReversiEnv, Player
    env = ReversiEnv().reset()
    env.step(19)                                           # black plays; white to move
    own, enemy = env.board.white, env.board.black
    ae = p.action_with_evaluation(own, enemy, callback_in_mtcs=CallbackInMCTS(10, cb))
    assert ae.action is not None and (find_correct_moves(own, enemy) >> ae.action) & 1
    assert len(calls) >= 3 and calls[0] > 0
    assert sum(p.var_n[ReversiPlayer.counter_key(ReversiEnv().update(own, enemy, Player.black))]) < 400
    assert len(p.moves) == 8 and p.ask_thought_about(own, enemy).action == ae.action


def test_evaluate_worker_batched_match(tmp_path):
    """The `eval` worker as one batched match (worker/evaluate.py of this package; reference: worker/evaluate.py:44-96):
    6 games of best model vs challenger all in flight on two engines == the same 6 games played one at a time by two
    ReversiPlayer objects and a ReversiEnv the way the reference's play_game does (same random streams); reproducible;
    the reference's file handling (challenger directory removed, best model replaced iff the winning rate reaches
    replace_rate) and its early-stopping decision rule."""
    import os
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.agent.model import ReversiModel
    from reversi_alpha_zero_amd.worker.evaluate import EvaluateWorker, decide
    from reversi_alpha_zero_amd.lib.model_helpler import save_as_best_model

    def make_cfg(root):
        cfg = Config()
        cfg.model.update(dict(cnn_filter_num=16, res_layer_num=1, value_fc_size=16))
        rc = cfg.resource
        rc.model_dir = str(root / "model")
        rc.model_best_config_path = str(root / "model" / "model_best_config.json")
        rc.model_best_weight_path = str(root / "model" / "model_best_weight.h5")
        rc.next_generation_model_dir = str(root / "model" / "next_generation")
        os.makedirs(rc.next_generation_model_dir)
        cfg.eval.game_num = 6
        cfg.eval.play_config.simulation_num_per_move = 8
        best = ReversiModel(cfg)
        best.build(seed=1)
        save_as_best_model(best)
        ng = ReversiModel(cfg)
        ng.build(seed=2)
        d = os.path.join(rc.next_generation_model_dir, rc.next_generation_model_dirname_tmpl % "20260101-000000.000000")
        os.makedirs(d)
        ng.save(os.path.join(d, rc.next_generation_model_config_filename), os.path.join(d, rc.next_generation_model_weight_filename))
        return cfg, best, ng

    outcomes = []
    for rep in range(2):
        root = tmp_path / f"run{rep}"
        os.makedirs(root)
        cfg, best, ng = make_cfg(root)
        w = EvaluateWorker(cfg, seed=11, device=DEV)
        w.best_model = w.load_best_model()
        ng_model, model_dir = w.load_next_generation_model()
        results = w.play_games(w.best_model, ng_model, 6)
        assert len(results) == 6 and {r[1] for r in results} == {True, False}    # both colour assignments occur
        for ng_win, best_is_black, (nb, nw) in results:
            assert ng_win in (0, 1, None) and 0 < nb + nw <= 64
        outcomes.append(results)
        if rep == 0:   # the batched games are the games the reference's sequential play_game would play on these streams
            for g in (0, 3, 5):
                assert w.play_game_sequential(w.best_model, ng_model, g, results[g][1]) == results[g], g
        assert w.start(max_models=1) == 1
        assert not os.path.exists(model_dir)                      # challenger consumed
        kept = ReversiModel(cfg)
        assert kept.load(cfg.resource.model_best_config_path, cfg.resource.model_best_weight_path)
        better, played, rate = decide([r[0] for r in w.last_results] + [None] * 6, 6, cfg.eval.replace_rate)
        assert kept.model.to_blob() == (ng.model.to_blob() if better else best.model.to_blob())
    assert outcomes[0] == outcomes[1]                             # reproducible for a seed
    # evaluate.py:55-60: stop as soon as the losses reach game_num * (1 - replace_rate) or the wins game_num * replace_rate
    assert decide([0, 0, 1, 0, 1, 1], 6, 0.55) == (False, 4, 0.25)
    assert decide([1, None, 1, 1, 1, 0], 6, 0.55) == (True, 5, 1.0)
    assert decide([None] * 6, 6, 0.55) == (False, 6, 0)


def test_evaluate_worker_matches_equal_the_reference_evaluate_games():
    """tests/golden/eval_games.json: games the UNMODIFIED reference's EvaluateWorker.play_game played (worker/evaluate.py:66-96 - two
    ReversiPlayers with trees of their own, best model against challenger, colours drawn by evaluate.py:71; generated by
    tests/golden/make_golden_eval.py) - replayed as batched matches on the device: every game's colour draw, every ply's mover,
    action (resignations included) and root visit counts, the outcome (ng_win) and the final disc counts are the reference's.  Three
    matches: evaluate.py as shipped (parallel_search_num 8 on the raz-sched-v1 schedule, root solver from turn 50 by play_config,
    in-simulation solver from turn 50 by config.play), one simulation in flight with an early resign threshold and no solver, and
    one whose two config sections DISAGREE where the reference's player reads the global one (virtual_loss,
    use_solver_turn_in_simulation: agent/player.py:237,264)."""
    import json
    import os
    import hashlib
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.worker.evaluate import EvaluateWorker
    from conftest import ROOT
    with open(os.path.join(ROOT, "tests", "golden", "eval_games.json")) as f:
        gold = json.load(f)
    nets = []
    for meta in gold["nets"]:
        net = ReversiNet(meta["filters"], meta["res_layers"], meta["value_fc"]).keras_init_(meta["keras_init_seed"]).randomize_bn_(meta["randomize_bn_seed"])
        assert hashlib.sha256(net.to_blob()).hexdigest() == meta["blob_sha256"], "seeded net init is not reproducible"
        nets.append(net)
    for m in gold["matches"]:
        cfg = Config()
        cfg.eval.play_config.update(m["resolved_play_config"])
        cfg.play.update(m["resolved_config_play"])     # the section agent/player.py:127,237,264,410 reads whatever play_config says
        w = EvaluateWorker(cfg, seed=m["seed"], device=DEV)
        w.games_played = m["first_game_id"]
        results = w.play_games(nets[0], nets[1], len(m["games"]))
        for g, (ref, got, res) in enumerate(zip(m["games"], w.last_games, results)):
            where = (m["name"], ref["game_id"])
            assert got["game_id"] == ref["game_id"] and got["best_is_black"] == ref["best_is_black"], where
            assert [(p["who"], p["action"]) for p in got["plies"]] == [(p["who"], p["action"]) for p in ref["plies"]], where
            for i, (a, b) in enumerate(zip(got["plies"], ref["plies"])):
                if b["root_n"] is not None:      # (a move the end-game solver decided has no search behind it)
                    want = [0.0] * 64
                    for k, v in b["root_n"].items():
                        want[int(k)] = v
                    assert [float(x) for x in a["root_n"]] == want, (where, i)
            assert res[0] == ref["ng_win"] and res[1] == ref["best_is_black"] and list(res[2]) == ref["black_white"], where


def test_training_tensors_on_device():
    """lib/data_helper.training_tensors: packed bitboards -> float planes by raz_planes_batch == the
    reference's bit_to_array recipe (worker/optimize.py:214-231)."""
    from reversi_alpha_zero_amd.lib.data_helper import convert_to_training_data, training_tensors
    rng = np.random.default_rng(5)
    rows = []
    for _ in range(257):
        own = int(rng.integers(0, 2**64, dtype=np.uint64))
        enemy = int(rng.integers(0, 2**64, dtype=np.uint64)) & ~own
        p = rng.random(64)
        rows.append([[own, enemy], list(p / p.sum()), int(rng.integers(-1, 2))])
    state, policy, z = training_tensors(rows, DEV)
    es, ep, ez = convert_to_training_data(rows)
    assert np.array_equal(state.cpu().numpy(), es.astype(np.float32))
    assert np.allclose(policy.cpu().numpy(), ep, atol=1e-7) and np.array_equal(z.cpu().numpy(), ez.astype(np.float32))
