"""ococc_sparse_conv_tile_bf16 (compact-then-multiply sub-manifold convolution for sparse active sets)
against the output-stationary kernels, which are pinned to the oracle in test_gpu_spconv.py
(reference: indiceConv / indiceConvBackward, mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-456)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _scene(dev, batch, shape, density, seed):
    g = torch.Generator().manual_seed(seed)
    cells = batch * shape[0] * shape[1] * shape[2]
    n = max(int(cells * density), 1)
    flat = torch.randperm(cells, generator=g)[:n].sort().values
    b = flat // (shape[0] * shape[1] * shape[2])
    r = flat % (shape[0] * shape[1] * shape[2])
    z, y, x = r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]
    return torch.stack([b, z, y, x], 1).to(torch.int32).to(dev)


@pytest.mark.parametrize('cin,cout', [(32, 64), (64, 32), (128, 64), (64, 128), (32, 32), (128, 128)])
@pytest.mark.parametrize('density', [0.03, 0.45])
def test_tile_kernel_matches_output_stationary_kernels(dev, cin, cout, density):
    from objectcentricocccompletion_amd.spconv import ops
    shape = [12, 10, 11]
    coors = _scene(dev, 5, shape, density, seed=cin + cout)
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 5, shape, 3, subm=True)
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.1).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    for dt, tol in ((torch.bfloat16, 2e-2), (torch.float32, 2e-5)):
        x = torch.randn(n, cin, generator=g).to(dev).to(dt)
        dy = torch.randn(n, cout, generator=g).to(dev).to(dt)
        res = {}
        for tile in (False, True):
            ops.SPARSE_TILE_CONV = tile
            try:
                y = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias)
                dx, dw = ops.indice_conv_backward(x, w, dy, pairs, num, False, True)
                y2 = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias)
            finally:
                ops.SPARSE_TILE_CONV = None
            assert torch.equal(y, y2)                                   # deterministic, run to run
            res[tile] = (y.float(), dx.float(), dw.float())
        for a, b in zip(res[False], res[True]):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= tol * max(float(a.abs().max()), 1e-6)


def test_tile_kernel_fixed_capacity_rows_and_ragged_tail(dev):
    """Rows with -1 coordinates (fixed-capacity padding) take part in nothing and produce the bias; the row
    count is not a multiple of the tile."""
    from objectcentricocccompletion_amd.spconv import ops
    shape = [9, 9, 9]
    coors = _scene(dev, 3, shape, 0.2, seed=7)
    pad = torch.full((77, 4), -1, dtype=torch.int32, device=dev)
    coors = torch.cat([coors, pad], 0)
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 3, shape, 3, subm=True)
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(3, 3, 3, 64, 64, generator=g) * 0.1).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    x = torch.randn(n, 64, generator=g).to(dev).bfloat16()
    ops.SPARSE_TILE_CONV = True
    try:
        y = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias).float()
    finally:
        ops.SPARSE_TILE_CONV = None
    ye = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias).float()
    assert float((y - ye).abs().max()) <= 2e-2 * float(ye.abs().max())
    assert torch.equal(y[-77:], bias.bfloat16().float().expand(77, 64))


def test_density_hint_selects_the_kernel(dev):
    from objectcentricocccompletion_amd.spconv import ops
    shape = [8, 8, 8]
    coors = _scene(dev, 2, shape, 0.05, seed=11)
    _, pairs, num = ops.get_indice_pairs(coors, 2, shape, 3, subm=True)
    rb = pairs._ococc
    assert not ops._use_tile_kernel(rb, 128, 64)                # no hint: the general kernels
    ops.set_rulebook_density(pairs, 1.8)
    assert ops._use_tile_kernel(rb, 128, 64) and not ops._use_tile_kernel(rb, 16, 32)
    ops.set_rulebook_density(pairs, 2.5)
    assert ops._use_tile_kernel(rb, 128, 64) and not ops._use_tile_kernel(rb, 64, 32)
    ops.set_rulebook_density(pairs, 11.0)
    assert not ops._use_tile_kernel(rb, 128, 64)
