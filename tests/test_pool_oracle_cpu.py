"""The oracle's sparse max pool (oracle.indice_maxpool / indice_maxpool_backward: the reference's CPU functor,
mmdet3d/ops/spconv/src/maxpool.cc:9-55, with its zero-initialised output, include/spconv/pool_ops.h:34) pinned on CPU:
equal to torch's dense max_pool3d where all inputs are positive, zero where all inputs are negative, and the backward
pass equal to autograd's on tie-free positive inputs."""
import numpy as np
import torch

from oracle import oracle as O


def _case(seed):
    rng = np.random.default_rng(seed)
    B, shape, c = 2, (6, 7, 8), 5
    mask = rng.random((B,) + shape) < 0.35
    idx = np.argwhere(mask).astype(np.int32)
    oshape = [(s + 2 - 3) // 2 + 1 for s in shape]
    eo, ep, en = O.conv_rulebook(idx, B, oshape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    return rng, B, shape, c, idx, eo, ep, en


def test_oracle_maxpool_equals_dense_pool_on_positive_inputs_and_clamps_negative_ones():
    rng, B, shape, c, idx, eo, ep, en = _case(3)
    x = (rng.random((len(idx), c)) + 0.1).astype(np.float32)
    dense = torch.zeros((B, c) + shape, requires_grad=True)
    with torch.no_grad():
        dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.from_numpy(x)
    yd = torch.nn.functional.max_pool3d(dense, 3, stride=2, padding=1)
    out = O.indice_maxpool(x, ep, en, len(eo))
    assert np.array_equal(out, yd[eo[:, 0], :, eo[:, 1], eo[:, 2], eo[:, 3]].detach().numpy())
    dy = rng.standard_normal(out.shape).astype(np.float32)
    g = torch.zeros_like(yd)
    g[eo[:, 0], :, eo[:, 1], eo[:, 2], eo[:, 3]] = torch.from_numpy(dy)
    yd.backward(g)
    din = O.indice_maxpool_backward(x, out, dy, ep, en)
    assert np.allclose(din, dense.grad[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy(), atol=1e-6)
    # all-negative inputs: the reference's output never leaves its zero start, and no input equals it
    neg = -x
    assert not O.indice_maxpool(neg, ep, en, len(eo)).any()
    assert not O.indice_maxpool_backward(neg, np.zeros_like(out), dy, ep, en).any()
