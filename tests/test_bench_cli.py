"""CPU: bench.py's command-line contract where no GPU is needed to check it - it must refuse loudly (never fall back to a
CPU path), `--gpus N` started without a launcher must try to become N ranks, and the spot-check bookkeeping must count
the root-expansion simulation."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e, timeout=300)


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks the behaviour on a box without a GPU")
def test_bench_refuses_without_a_gpu():
    r = _run()
    assert r.returncode != 0 and "needs a GPU" in (r.stdout + r.stderr) and "no CPU fallback" in (r.stdout + r.stderr)


@pytest.mark.skipif(not _no_gpu(), reason="checks the behaviour on a box without a GPU")
def test_bench_gpus_n_spawns_or_refuses():
    r = _run("--gpus", "2")
    assert r.returncode != 0 and "only 0 GPU(s) visible" in (r.stdout + r.stderr)
    # a launcher's environment that disagrees with --gpus is an error, not a silent 1-GPU run
    r = _run("--gpus", "4", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_bench_line_schema_is_documented():
    """The keys the driver's contract names are assembled in headline_leg (a static check: no GPU here)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"parity_spotcheck"'):
        assert key in src, key
    assert "BASELINE configs[2]" in src and "oracle" in src.split("def headline_leg")[0]   # the oracle only in the checker functions
