import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def synth_input(name, shape, lo=-1.0, hi=1.0):
    """Same recipe as tools/make_golden.py: hash-seeded uniform input, a pure function of `name`."""
    from oracle.weightgen import uniform01
    u = uniform01("input:" + name, int(np.prod(shape)))
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def max_abs(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return (a - b).abs().max().item()
