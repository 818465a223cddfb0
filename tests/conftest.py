import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """libpayne_hip.so is built (or rebuilt, when a source is newer) before any test loads it: a clean checkout
    can run `pytest -m gpu` without a separate build step.  hipcc cross-compiles, so this also runs without a GPU."""
    from thepayne_amd.build import build_lib
    return build_lib()
