import importlib
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
def pc3d():
    """The product package (directory name is not an identifier, hence importlib)."""
    return importlib.import_module("3dpointcloudattack_amd")


@pytest.fixture(scope="session")
def ops(pc3d):
    return importlib.import_module("3dpointcloudattack_amd.ops")


@pytest.fixture(scope="session")
def metrics_fx():
    return np.load(os.path.join(GOLDEN, "metrics.npz"))


@pytest.fixture(scope="session")
def metrics4096_fx():
    return np.load(os.path.join(GOLDEN, "metrics_n4096.npz"))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
