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


def pytest_collection_finish(session):
    """GPU sessions: bring torch's HIP runtime up BEFORE the first libdynhip context exists.  The torch
    wheel carries its own copy of the ROCm runtime; initialised late -- after a dozen contexts of the
    system runtime that libdynhip.so links -- it reported "No HIP GPUs are available" on the MI355X box
    (tests/test_gpu_rccl.py needs torch.cuda for its RCCL process group; bench.py has always initialised
    torch first)."""
    if not any(item.get_closest_marker("gpu") for item in session.items):
        return
    if os.environ.get("DH_TEST_NO_TORCH"):  # sanitizer runs (make asan): the system runtime only
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    except Exception:  # no torch / no device: the gpu tests themselves will say so
        pass


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
