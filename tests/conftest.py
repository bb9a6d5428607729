import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")
    config.addinivalue_line(
        "markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_bounding():
    return np.load(os.path.join(GOLDEN, "bounding.npz"))


@pytest.fixture(scope="session")
def golden_proposals():
    return np.load(os.path.join(GOLDEN, "proposals.npz"))


@pytest.fixture(scope="session")
def golden_rng():
    return np.load(os.path.join(GOLDEN, "rng.npz"))


@pytest.fixture(scope="session")
def golden_runs():
    return np.load(os.path.join(GOLDEN, "runs.npz"))


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
