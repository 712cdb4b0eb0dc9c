import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class _G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return _G()


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_present():
    """A fresh checkout has no built artefacts (they are git-ignored): build the product library (hipcc cross-compiles
    gfx950 without a GPU) and the oracle's C restatement once, only when they are missing."""
    from pyaudiorestoration_amd import build as product_build
    if not os.path.exists(product_build.OUT):
        product_build.build(force=False, verbose=False)
    from oracle import oracle_c
    oracle_c.build()
