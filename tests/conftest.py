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
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP C-ABI library; GPU tests fail loudly (not skip) if it is missing on a GPU box."""
    import torch
    assert torch.cuda.is_available(), "GPU test collected without a GPU"
    from xpoint_amd import _lib
    return _lib.load()
