import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


import pytest


@pytest.fixture(autouse=True)
def _grad_mode_is_per_test():
    """torch's grad mode is process-global and several harness tests switch it (the reference's main() runs under torch.set_grad_enabled(False) and
    enables it around training): every test starts with gradients ENABLED and whatever it sets is undone afterwards."""
    import torch
    prev = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    yield
    torch.set_grad_enabled(prev)
