import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _seed_every_test():
    """Every test starts from the same torch RNG state (CPU and GPU): a failure reproduces, a pass is not luck.  Tests that want
    several draws loop over explicit seeds themselves."""
    import torch
    torch.manual_seed(20260927)
    yield
