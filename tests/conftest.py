import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import inclusivegan_amd  # noqa: E402,F401 -- before anything touches the GPU: the package sets the HIP runtime flags it needs (see its __init__)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no ROCm device')
    return torch.device('cuda', 0)


@pytest.fixture(autouse=True)
def _seed_torch():
    """Every test starts from the same host and device RNG state: the parity tests record the device draws of the
    HIP path and replay them into the oracle, so an unseeded run compares a different random problem each time."""
    import torch
    torch.manual_seed(20261002)
    yield
