"""CPU suite: the oracle's rulebook restatement (oracle/ococc_oracle.c) against vectors produced by
the REFERENCE's own geometry.h generators (tests/golden/rulebook.npz, oracle/gen_golden_rulebook.py,
oracle/_ref/rulebook_ref.so).  When the compiled reference itself is present (build container, and
the GPU box, where oracle/_ref travels as a prebuilt file) it is additionally called live on fresh
random inputs."""
import ctypes
import os

import numpy as np
import pytest

from oracle import oracle as O

SUBM = ['bench40', 'bench80', 'shuffled', 'dilated', 'k133']
CONV = ['down_k3s2p1', 'down_k2s2p0', 'down_aniso', 's1p0', 'up_k3s2p1', 'up_k2s2p0']


@pytest.fixture(scope='module')
def rb(golden_dir):
    return np.load(os.path.join(golden_dir, 'rulebook.npz'))


def expand(g, name, n):
    """compacted golden -> the reference's [K,2,N] buffer with its -1 fill."""
    num = g[name + '_num']
    pairs = np.full((len(num), 2, n), -1, np.int32)
    o = 0
    for k, c in enumerate(num):
        pairs[k, 0, :c] = g[name + '_pairs_in'][o:o + c]
        pairs[k, 1, :c] = g[name + '_pairs_out'][o:o + c]
        o += c
    return pairs, num


@pytest.mark.parametrize('name', SUBM)
def test_oracle_subm_rulebook_equals_reference(rb, name):
    idx = rb[name + '_indices']
    pairs, num = O.subm_rulebook(idx, int(rb[name + '_batch']), rb[name + '_shape'], rb[name + '_ksize'],
                                 rb[name + '_dilation'])
    ep, en = expand(rb, name, len(idx))
    assert np.array_equal(num, en)
    assert np.array_equal(pairs, ep)


@pytest.mark.parametrize('name', CONV)
def test_oracle_conv_rulebook_equals_reference(rb, name):
    idx = rb[name + '_indices']
    outi, pairs, num = O.conv_rulebook(idx, int(rb[name + '_batch']), rb[name + '_out_shape'], rb[name + '_ksize'],
                                       rb[name + '_stride'], rb[name + '_padding'], rb[name + '_dilation'],
                                       transpose=bool(rb[name + '_transpose']))
    ep, en = expand(rb, name, len(idx))
    assert np.array_equal(outi, rb[name + '_out_indices'])
    assert np.array_equal(num, en)
    assert np.array_equal(pairs, ep)


def test_oracle_indice_conv_equals_reference_functor_loop(rb):
    # expected: the reference's CPU gather / scatter-add functors (src/reordering.cc:21-50) in the
    # indiceConv loop of spconv_ops.h:300-354
    idx = rb['shuffled_indices']
    pairs, num = O.subm_rulebook(idx, 3, rb['shuffled_shape'])
    y = O.indice_conv(rb['functor_x'], rb['functor_w'], pairs, num, len(idx), subm=True)
    assert np.allclose(y, rb['functor_y'], rtol=1e-5, atol=1e-5)


_REF = os.path.join(os.path.dirname(os.path.abspath(O.__file__)), '_ref', 'rulebook_ref.so')


@pytest.mark.skipif(not os.path.exists(_REF), reason='compiled reference (oracle/_ref) not present')
@pytest.mark.parametrize('seed', range(4))
def test_oracle_equals_compiled_reference_live(seed):
    from oracle.gen_golden_rulebook import load_ref, ref_conv, ref_subm
    lib = load_ref()
    rng = np.random.default_rng(100 + seed)
    B = int(rng.integers(1, 4))
    shape = [int(v) for v in rng.integers(3, 14, 3)]
    idx = np.argwhere(rng.random([B] + shape) < rng.uniform(0.05, 0.7)).astype(np.int32)
    idx = idx[rng.permutation(len(idx))]
    ks = [int(v) for v in rng.choice([1, 2, 3], 3)]
    ep, en = ref_subm(lib, idx, B, shape, [k | 1 for k in ks])
    p, n = O.subm_rulebook(idx, B, shape, [k | 1 for k in ks])
    assert np.array_equal(n, en) and np.array_equal(p, ep)
    st = [int(v) for v in rng.integers(1, 3, 3)]
    pad = [int(v) for v in rng.integers(0, 2, 3)]
    for tr in (False, True):
        if tr:
            osh = [(shape[i] - 1) * st[i] - 2 * pad[i] + ks[i] for i in range(3)]
        else:
            osh = [(shape[i] + 2 * pad[i] - (ks[i] - 1) - 1) // st[i] + 1 for i in range(3)]
        if min(osh) < 1:
            continue
        eo, ep, en = ref_conv(lib, idx, B, osh, ks, st, pad, [1, 1, 1], tr)
        o, p, n = O.conv_rulebook(idx, B, osh, ks, st, pad, [1, 1, 1], transpose=tr)
        assert np.array_equal(o, eo) and np.array_equal(n, en) and np.array_equal(p, ep)
