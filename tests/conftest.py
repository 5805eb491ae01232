import os

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')  # see objectcentricocccompletion_amd/graph.py
# The graph-pair replay of the temporal transformer / head tail (heads.graphed_call) is exercised by its own tests in a
# CHILD process (tests/test_gpu_ococc_train.py::test_graph_pair_paths_in_a_child_process): one segmentation fault inside
# hipGraphLaunch (backward graph replayed from autograd's device thread) was seen in ~60 runs of the suite and not again
# in 40 targeted repeats; every other test runs those modules eagerly, so that such a fault cannot take the suite down.
os.environ.setdefault('OCOCC_GRAPH_TRANSFORMER', '0')

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
