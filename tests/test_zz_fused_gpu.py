"""GPU: the fused tree + net kernel (csrc/raz_engine_fused.hip k_tree_net, raz_engine_config.reserved bit 4 - opt-in) on the device:
a batch stepped by it must leave exactly the records the classic k_tree + k_net_mfma pipeline leaves (bit for bit: same
operations per game in the same order), and sampled games must equal the CPU oracle.

The kernel was developed on the wave emulator (tests/test_engine_fused_emu.py: reference golden game, oracle batches, pruning,
solver, series, continuous batching - all bit-exact on the CPU) in a round whose GPU minutes were spent; this file is its first
hardware run.  It therefore runs in a child process (whatever it does cannot take the test session with it) and is marked
xfail(strict=False): a pass is reported as XPASS, a failure as xfailed - the outcome is in the pytest summary either way and
nothing else depends on this kernel.  The file name sorts last on purpose."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

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
for variant, solver in (("mini_shared", False), ("agz", False), ("mini_solver_noresign", True)):
    cfg = config_of(next(g for g in gold["games"] if g["variant"] == variant))
    n, sims = (96, 14) if not solver else (8, 10)
    recs = []
    for fused in (False, True):
        eng = SelfPlayEngine(cfg, DeviceNet(blob, "cuda:0"), n_games=n, seed=5, sims_hint=sims, record_root_w=True, fused=fused)
        eng.start(1000, sims)
        eng.run(chunk=37)       # launches end in the middle of searches
        recs.append(eng.records(save_policy_of_tau_1=True))
        del eng
    assert recs[0] == recs[1], variant + ": fused records differ from the classic pipeline's"
    ocfg = O.play_cfg_from_config(cfg)
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
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and "FUSED_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
