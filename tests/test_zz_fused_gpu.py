"""GPU: the fused tree + net kernel (csrc/raz_engine_fused.hip k_tree_net / k_tree_par_net, raz_engine_config.reserved bit 4 - what
BatchedSelfPlayWorker runs for 16-filter nets) on the device: a batch stepped by it must leave exactly the records the two-kernel
pipeline k_tree + k_net_mfma leaves (bit for bit: same operations per game in the same order), and sampled games must equal the CPU
oracle.  Developed on the wave emulator (tests/test_engine_fused_emu.py: reference golden game, oracle batches, pruning, solver,
series, continuous batching - all bit-exact on the CPU); first hardware run in round 3's round-end suite (passed).  The comparison
runs in a child process with a timeout, so that a kernel that hangs costs this test, not the session."""
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
