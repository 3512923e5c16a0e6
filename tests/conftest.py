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


# Tests that assert BIT-equality of runs (the default, deterministic mode's contract). Under PC3D_DETERMINISTIC=0 the
# float-atomic kernels are selected, for which those assertions do not hold by design: they are skipped there, and
# everything else — values, gradients, goldens at their tolerances — runs in both modes.
_BITWISE_ONLY = ("test_determinism_gpu.py", "test_configs_gpu.py::test_geoa3_on_", "test_configs_gpu.py::test_cw_on_curvenet",
                 "test_configs_gpu.py::test_victim_input_knn_is_shared", "test_wt_cache_eviction_keeps_captured_graphs_valid",
                 "direct_terms_vs_general_form")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("PC3D_DETERMINISTIC", "1") != "0":
        return
    skip = pytest.mark.skip(reason="asserts bit-equality: deterministic mode only (PC3D_DETERMINISTIC=0 is set)")
    for it in items:
        if any(k in it.nodeid for k in _BITWISE_ONLY):
            it.add_marker(skip)


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
