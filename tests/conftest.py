"""pytest configuration: markers, paths, shared fixtures."""
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
    # GPU runs use torch (streams, slab buffers) next to libsph_hip.so: torch's bundled HIP runtime must be
    # the one that gets loaded, i.e. torch first (gpufluidsimulator_amd.capi.load explains why)
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        import torch  # noqa: F401


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def bits(a):
    """View float32 data as uint32 for bit-exact comparisons."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
