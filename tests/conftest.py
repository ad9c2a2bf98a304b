import importlib
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
def pkg():
    """The product package (hyphenated directory name -> importlib)."""
    return importlib.import_module("moving-mnist-vae_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import vae_oracle
    return vae_oracle


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
