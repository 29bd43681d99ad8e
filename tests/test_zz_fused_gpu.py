"""GPU: the fused tree + net kernel (csrc/raz_engine_fused.hip k_tree_net, raz_engine_config.reserved bit 4 - opt-in) on the device:
a batch stepped by it must leave exactly the records the classic k_tree + k_net_mfma pipeline leaves (bit for bit: same
operations per game in the same order), and sampled games must equal the CPU oracle.

The kernel was developed on the wave emulator (tests/test_engine_fused_emu.py: reference golden game, oracle batches, pruning,
solver, series, continuous batching - all bit-exact on the CPU) in a round whose GPU minutes were spent; this file is its first
hardware run.  It therefore runs in a child process (whatever it does cannot take the test session with it) and is marked
xfail(strict=False): a pass is reported as XPASS, a failure as xfailed - the outcome is in the pytest summary either way and
nothing else depends on this kernel.  The file name sorts last on purpose.

Also here, after it, for the same reason (the driver runs the suite with -x; nothing may stand behind a test whose exact form has
not run on hardware yet): the bit-equality test of the narrow-net kernel's two- / four-waves-per-position variants - the kernels
themselves were checked on the device this round (profiles/r3/quick_split_positions_over_waves.log), this pytest form of the
check was written afterwards."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

DEV = "cuda:0"

pytestmark = pytest.mark.gpu

_CHILD = '''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import numpy as np, torch
import oracle as O
from oracle_util import load_mcts_golden, golden_net_blob, config_of
from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine
gold = load_mcts_golden()
blob = golden_net_blob(gold["net"])
out = {{}}
for variant, solver, par in (("mini_shared", False, 1), ("agz", False, 1), ("mini_solver_noresign", True, 1), ("mini_shared", False, 4)):
    cfg = config_of(next(g for g in gold["games"] if g["variant"] == variant))
    if par > 1:     # k_tree_par_net: simulation slots on the raz-sched-v1 rounds
        cfg.play.parallel_search_num, cfg.play.thinking_loop = par, 1
    n, sims = (96, 14) if not solver else (8, 10)
    recs = []
    for fused in (False, True):
        eng = SelfPlayEngine(cfg, DeviceNet(blob, "cuda:0"), n_games=n, seed=5, sims_hint=sims, record_root_w=True, fused=fused)
        eng.start(1000, sims)
        eng.run(chunk=37)       # launches end in the middle of searches
        recs.append(eng.records(save_policy_of_tau_1=True))
        del eng
    assert recs[0] == recs[1], variant + ": fused records differ from the classic pipeline's"
    ocfg = O.play_cfg_from_config(cfg, parallel_search_num=par) if par > 1 else O.play_cfg_from_config(cfg)
    for i in (0, n - 1):
        plies, summ = O.selfplay_game(ocfg, blob, 5, 1000 + i, sims)
        got_plies, got_sum = recs[1][i]
        assert len(plies) == len(got_plies) and got_sum["winner"] == summ["winner"], (variant, i)
        for a, b in zip(got_plies, plies):
            assert a["action"] == b["action"] and a["root_n"] == b["root_n"] and a["root_w"] == b["root_w"], (variant, i)
print("FUSED_OK")
'''


@pytest.mark.xfail(reason="first hardware run of the opt-in fused kernel (validated on the wave emulator only so far)", strict=False)
def test_fused_kernel_equals_the_classic_pipeline_and_the_oracle_on_the_device():
    p = subprocess.Popen([sys.executable, "-c", _CHILD.format(root=ROOT)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=dict(os.environ, PYTHONPATH=ROOT))
    try:
        so, se = p.communicate(timeout=420)
    except subprocess.TimeoutExpired:
        p.kill()
        try:
            p.communicate(timeout=15)
        except subprocess.TimeoutExpired:
            pass    # stuck in the driver: abandoned, not waited for
        pytest.fail("the child process did not finish within 420 s")
    assert p.returncode == 0 and "FUSED_OK" in so, (p.returncode, so[-1500:], se[-3000:])

def _positions(n, seed):
    rng = np.random.default_rng(seed)
    own = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    return own, rng.integers(0, 2**64, size=n, dtype=np.uint64) & ~own


@pytest.mark.parametrize("shape", [(16, 1, 16), (16, 2, 48), (16, 3, 16)])
def test_net_split_kernels_equal_other_kernels(shape):
    """k_net_mfma16_split (one position on a workgroup of two / four waves, the two heads side by side; an opt-in latency
    variant, see its header) == k_net_mfma (one wave per position) == k_net_wave (VALU), bit for bit, on a ragged batch
    with an active mask; R != 1 takes the form whose conv operands are fetched per layer."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet
    from reversi_alpha_zero_amd.engine import DeviceNet
    blob = ReversiNet(*shape).keras_init_(2).randomize_bn_(3).to_blob()
    n = 4101
    own, enemy = _positions(n, 7)
    o, e = torch.from_numpy(own.view(np.int64)).to(DEV), torch.from_numpy(enemy.view(np.int64)).to(DEV)
    act = torch.from_numpy((np.random.default_rng(8).random(n) < 0.8).astype(np.uint8)).to(DEV)
    outs = []
    for kernel in ("mfma_split2", "mfma_split4", None, "mfma_wave", "valu"):
        p, v = DeviceNet(blob, DEV, kernel=kernel).predict_bitboards(o, e, active=act)
        outs.append((p.view(torch.int32), v.view(torch.int32)))
    for p, v in outs[1:]:
        assert torch.equal(outs[0][0], p) and torch.equal(outs[0][1], v)
    assert bool((outs[0][0][act == 0] == 0).all()) and bool((outs[0][0][act == 1] != 0).any())
    p_all, v_all = DeviceNet(blob, DEV, kernel="mfma_split4").predict_bitboards(o[:9], e[:9])   # a handful of positions
    p_ref, v_ref = DeviceNet(blob, DEV, kernel="valu").predict_bitboards(o[:9], e[:9])
    assert torch.equal(p_all.view(torch.int32), p_ref.view(torch.int32)) and torch.equal(v_all.view(torch.int32), v_ref.view(torch.int32))
