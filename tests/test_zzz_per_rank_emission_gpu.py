"""GPU: per-rank emission (BatchedSelfPlayWorker.run, emission "auto" with a rank's id range a whole number of files) on the real
engine - two ranks share cuda:0 and rendezvous over gloo, each copies its OWN packed records from HBM and writes its own files,
only the 32-byte summaries are gathered.  (Written when no GPU minutes seemed left, hence sorted last - so that the tests of
record before it would run whatever happened here; it has since passed on an MI355X:
profiles/r5/pytest_gpu_per_rank_emission_two_ranks_on_one_gpu.log.  CPU coverage of the same path: tests/test_worker_emu.py on
the wave emulator, tests/test_worker_run_host.py on a stub engine.)"""
import re

import pytest

import test_multirank_gpu as M

pytestmark = pytest.mark.gpu


def test_two_ranks_writing_their_own_files_equal_one_rank(tmp_path, monkeypatch):
    """2 ranks x 10 ids x 2 blocks, 5 games per play file and per GGF file (2 files per rank and block) vs 1 rank x 20 ids x 2
    blocks: play_*.json byte-identical in name order, GGF files identical (date aside), game index 40, both ranks end under the
    one-rank run's threshold (stepped on rank 0 after the first block and broadcast)."""
    script = M._WORKER_SCRIPT.replace("nb_game_in_ggf_file=7", "nb_game_in_ggf_file=5")
    assert "nb_game_in_ggf_file=5" in script and "nb_game_in_file=5" in script   # 10 ids per rank = 2 whole files: "auto" = per rank
    monkeypatch.setattr(M, "_WORKER_SCRIPT", script)
    one = M._run(tmp_path, "one", 1, 20, 40)
    two = M._run(tmp_path, "two", 2, 10, 40)
    norm = lambda texts: [re.sub(r"DT\[[^\]]*\]", "DT[]", t) for t in texts]
    assert len(one[0]) >= 6 and one[0] == two[0]
    assert len(one[1]) >= 6 and norm(one[1]) == norm(two[1])
    assert one[2] == two[2] == "40"
    t1 = {line.split()[2] for line in one[3]}
    t2 = {line.split()[2] for line in two[3]}
    assert len(two[3]) == 2 and len(t2) == 1 and t1 == t2, (one[3], two[3])
