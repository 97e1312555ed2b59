import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """A fresh checkout has no libpdeopt_hip.so (built artefacts are git-ignored): build it once -- hipcc
    cross-compiles for gfx950 without a GPU; nothing happens when the library is up to date.  The C oracle
    (test infrastructure) builds itself on first use."""
    lib = os.path.join(ROOT, "pde_opt_amd", "libpdeopt_hip.so")
    if not os.path.exists(lib):
        from pde_opt_amd.csrc import build as B

        B.build(verbose=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))

    return load
