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
