"""GPU parity of the compact-multiply-pull convolution (csrc/sparse_conv_pull.hip) against the oracle and against the
output-stationary kernels, sparse and dense neighbourhoods, ragged tails, f32 and bf16 outputs."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from test_gpu_spconv import TOL, _conv_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('density', [0.02, 0.04, 0.3, 0.9])
@pytest.mark.parametrize('shape', [(14, 15, 16), (40, 40, 40)])
def test_pull_kernel_forward_vs_oracle_and_stream_kernel(dev, density, shape):
    from objectcentricocccompletion_amd.spconv import ops
    if shape[0] == 40 and density > 0.1:
        pytest.skip('the oracle takes minutes on dense 40^3 grids')
    rng = np.random.default_rng(int(density * 100) + shape[0])
    idx, x, w, dy, pairs, num, ep, en = _conv_case(rng, dev, 3, shape, density, 64, 128)
    n = len(idx)
    xt, wt = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    outs = {}
    for pull in (True, False):
        ops.PULL_CONV = pull
        try:
            outs[pull] = (ops.indice_conv(xt, wt, pairs, num, n, False, True),
                          ops.indice_conv(xt.bfloat16(), wt.bfloat16(), pairs, num, n, False, True))
        finally:
            ops.PULL_CONV = None
    ey = O.indice_conv(x, w, ep, en, n, subm=True)
    y, yb = outs[True]
    assert y.dtype == torch.float32 and np.allclose(y.cpu().numpy(), ey, **TOL)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb, y.bfloat16())   # bf16 output = RNE rounding of the f32 sums
    # same products, same order per row (centre first, then ascending offsets) as the output-stationary kernel
    assert np.allclose(y.cpu().numpy(), outs[False][0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    y2 = None
    ops.PULL_CONV = True
    try:
        y2 = ops.indice_conv(xt, wt, pairs, num, n, False, True)
    finally:
        ops.PULL_CONV = None
    assert torch.equal(y, y2)   # deterministic


def test_pull_kernel_is_opt_in(dev):
    from objectcentricocccompletion_amd.spconv import ops
    rng = np.random.default_rng(1)
    idx, x, w, dy, pairs, num, ep, en = _conv_case(rng, dev, 2, (14, 15, 16), 0.04, 64, 128)
    rb = pairs._ococc
    assert not ops._use_pull_kernel(rb, 64, 128)          # no hint: the general kernels
    ops.set_rulebook_density(pairs, 1.7)
    assert not ops._use_pull_kernel(rb, 64, 128)          # measured slower than the streamed-weights kernel: opt-in only
    ops._PULL_SHAPES[(64, 128)] = 3.0
    try:
        assert ops._use_pull_kernel(rb, 64, 128) and ops._fragment_major(rb, 64, 128) and not ops._use_pull_kernel(rb, 128, 128)
        ops.set_rulebook_density(pairs, 9.0)
        assert not ops._use_pull_kernel(rb, 64, 128)
    finally:
        del ops._PULL_SHAPES[(64, 128)]
