"""pytest configuration: markers, import paths, shared fixtures.

  -m "not gpu": oracle vs golden vectors, host logic, C-ABI symbol/loader checks (CPU only).
  -m gpu      : the parity tests proper — HIP path (through the C ABI) vs the oracle, on an MI355X.
`needs_reference` tests import the unmodified reference from /root/reference and are skipped
automatically where that tree does not exist (e.g. on the GPU box).
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") is eleven minutes of single-threaded work - whole games on the wave emulator, the unmodified
    reference played against the oracle - in tests that share nothing but read-only fixtures (the two-rank gloo tests have a port
    each): run it on worker processes (pytest-xdist) unless the caller chose a count himself or set RAZ_TESTS_SERIAL.  The GPU
    suite stays in one process: one GPU, and the loader of the HIP library is what its run records."""
    opt = config.option
    # never inside a worker process: a worker that starts workers of its own is a fork bomb (xdist sets PYTEST_XDIST_WORKER in
    # its workers; the sentinel below is inherited by everything this process starts, whatever xdist does)
    if os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("RAZ_TESTS_PARENT") or hasattr(config, "workerinput"):
        return None
    if (getattr(opt, "markexpr", "") == "not gpu" and getattr(opt, "numprocesses", 0) is None and not os.environ.get("RAZ_TESTS_SERIAL")
            and config.pluginmanager.hasplugin("xdist")):
        os.environ["RAZ_TESTS_PARENT"] = str(os.getpid())
        opt.numprocesses = max(1, min(6, (os.cpu_count() or 2) - 2))   # (before pytest-xdist's own hook reads it; set here too in case the order changes)
        opt.dist, opt.tx = "load", ["popen"] * opt.numprocesses


def _locked(name):
    """A lock file under tests/native: the worker processes build the shared libraries one at a time."""
    import contextlib
    import fcntl

    @contextlib.contextmanager
    def cm():
        os.makedirs(os.path.join(ROOT, "tests", "native"), exist_ok=True)
        with open(os.path.join(ROOT, "tests", "native", f".{name}.lock"), "w") as f:
            fcntl.flock(f, fcntl.LOCK_EX)
            try:
                yield
            finally:
                fcntl.flock(f, fcntl.LOCK_UN)
    return cm()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "needs_reference: imports /root/reference (skipped if absent)")


def pytest_collection_modifyitems(config, items):
    import ref_harness
    have_ref = ref_harness.reference_available()
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure libraz.so and liboracle.so exist (the .so files travel with the snapshot to the GPU
    box; here they are rebuilt only when sources are newer)."""
    import __graft_entry__ as g
    with _locked("build"):
        g.build()


@pytest.fixture(scope="session")
def golden_bb():
    with open(os.path.join(GOLDEN, "bitboard_env.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.load()


def H(s):
    return int(s, 16)
