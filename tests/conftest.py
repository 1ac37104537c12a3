import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


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
