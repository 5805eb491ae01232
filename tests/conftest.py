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


_SWITCH_DEFAULTS = {}   # module name -> {switch: value at import}: restored in front of every test
_SWITCHES = {
    'objectcentricocccompletion_amd.spconv.ops': ('SORTED_CONV', 'SPARSE_TILE_CONV', 'DEFAULT_PAIRS_PER_ROW', 'AUTO_DENSITY',
                                                  'FUSE_LN_BACKWARD', 'SORTED_TILES', '_TILE_SHAPES', 'SORTED_CONV_LN'),
    'objectcentricocccompletion_amd.spconv.modules': ('FUSE_CONV_LN', 'FUSE_TILE_CONV_LN'),
}


@pytest.fixture(autouse=True)
def _fresh_density_estimate():
    """the rulebook density estimate (spconv.ops.density: what the kernel choice keys on) is process-wide and lags one
    build behind the data: a test must not inherit the previous test's grids -- a stale estimate flips the kernel family
    between two passes of the same test (every family is correct, they are not bit-identical to each other).  The same for
    the module-level kernel switches a test may have left set (a helper of test_gpu_deferred.py did): every test starts
    from the values the modules were imported with."""
    for name, switches in _SWITCHES.items():
        mod = sys.modules.get(name)
        if mod is None:   # (imported here so that the defaults are taken before any test has touched them)
            try:
                mod = __import__(name, fromlist=['_'])
            except Exception:   # noqa: BLE001 -- the library is not built: the tests that need it say so themselves
                continue
        saved = _SWITCH_DEFAULTS.setdefault(name, {k: getattr(mod, k) for k in switches})
        for k, v in saved.items():
            setattr(mod, k, v)
    ops = sys.modules.get('objectcentricocccompletion_amd.spconv.ops')
    if ops is not None:
        ops.density.reset()
    yield
