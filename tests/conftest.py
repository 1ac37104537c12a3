import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CPU checkers (oracle/, oracle/_ref) are OpenMP code and most tests call them on tiny stripes.  A team as wide
# as the host (128+ threads on the GPU boxes) makes every such call pay for a full-machine barrier, and when the
# box's CPUs are shared or quota-limited that cost explodes (seconds per call).  A moderate, fixed team keeps the
# suite's run time predictable; it has to be set before libgomp is loaded.
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference behind oracle/_ref (prebuilt; skipped when absent)."""
    from oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built (no /root/reference at build time)")
    return Reference()


@pytest.fixture(scope="session")
def golden_hashes():
    with open(os.path.join(GOLDEN, "golden_hashes.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_vectors():
    return dict(np.load(os.path.join(GOLDEN, "golden_vectors.npz")))


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if stale) and loads libfastecc_hip.so."""
    import __graft_entry__ as ge
    ge.build()
    import fastecc_amd
    return fastecc_amd.lib()
