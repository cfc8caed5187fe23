import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_attention():
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "attention.pt"), weights_only=True)


@pytest.fixture(scope="session")
def golden_quant():
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "quant.pt"), weights_only=True)


@pytest.fixture(scope="session")
def golden_elementwise():
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "elementwise.pt"), weights_only=True)


@pytest.fixture(scope="session")
def golden_sched():
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "sched.pt"), weights_only=True)
