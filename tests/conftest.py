import os

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')  # see objectcentricocccompletion_amd/graph.py

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no ROCm device')
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _fresh_density_estimate():
    """the rulebook density estimate (spconv.ops.density: what the kernel choice keys on) is process-wide and lags one
    build behind the data: a test must not inherit the previous test's grids -- a stale estimate flips the kernel family
    between two passes of the same test (every family is correct, they are not bit-identical to each other)"""
    ops = sys.modules.get('objectcentricocccompletion_amd.spconv.ops')
    if ops is not None:
        ops.density.reset()
    yield
