import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def params_1k():
    from fpyv_amd import load_params
    return load_params(fps=1000)


@pytest.fixture(scope="session")
def params_60():
    from fpyv_amd import load_params
    return load_params()


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): compile the HIP library
    (hipcc cross-compiles gfx950 without a GPU) and the CPU checker once, before collection."""
    lib = os.path.join(REPO, "fpyv_amd", "libfpv_hip.so")
    chk = os.path.join(REPO, "oracle", "_build", "libfpv_lane_model.so")
    if not (os.path.isfile(lib) and os.path.isfile(chk)):
        import __graft_entry__
        __graft_entry__.build()
