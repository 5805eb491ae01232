"""GPU parity, voxel side (B1, B2, A5): HIP kernels through the C ABI vs the oracle, the
reference's golden vectors, and size-independent properties at benchmark scale."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vox(golden_dir):
    return np.load(os.path.join(golden_dir, 'voxelize.npz'))


@pytest.mark.parametrize('name', ['grid40', 'box', 'kitti'])
def test_dynamic_voxelize_golden(dev, vox, name):
    from objectcentricocccompletion_amd.voxel import voxelization
    pts = torch.from_numpy(vox[name + '_points']).to(dev)
    coors = voxelization(pts, vox[name + '_voxel_size'].tolist(), vox[name + '_range'].tolist(), -1, -1)
    assert coors.dtype == torch.int32
    assert np.array_equal(coors.cpu().numpy(), vox[name + '_dyn_coors'])  # bit exact


@pytest.mark.parametrize('name', ['grid40', 'box', 'kitti'])
@pytest.mark.parametrize('caps', [(5, 300), (35, 20000), (1, 7)])
def test_hard_voxelize_golden(dev, vox, name, caps):
    from objectcentricocccompletion_amd.voxel import voxelization
    mp, mv = caps
    pts = torch.from_numpy(vox[name + '_points']).to(dev)
    v, c, n = voxelization(pts, vox[name + '_voxel_size'].tolist(), vox[name + '_range'].tolist(), mp, mv)
    key = f'{name}_hard_{mp}_{mv}'
    assert np.array_equal(c.cpu().numpy(), vox[key + '_coors'])
    assert np.array_equal(n.cpu().numpy(), vox[key + '_npv'])
    assert np.array_equal(v.cpu().numpy(), vox[key + '_voxels'])


def test_dynamic_voxelize_large_vs_oracle(dev):
    from objectcentricocccompletion_amd.voxel import voxelization
    g = torch.Generator().manual_seed(7)
    pts = (torch.rand(1_000_003, 6, generator=g) * 2 - 1) * 4.7
    coors = voxelization(pts.to(dev), [0.2, 0.2, 0.2], [-4, -4, -4, 4, 4, 4], -1, -1)
    exp = O.dynamic_voxelize(pts.numpy(), [0.2, 0.2, 0.2], [-4, -4, -4, 4, 4, 4])
    assert np.array_equal(coors.cpu().numpy(), exp)
    empty = voxelization(torch.zeros(0, 4, device=dev), [0.2, 0.2, 0.2], [-4, -4, -4, 4, 4, 4], -1, -1)
    assert tuple(empty.shape) == (0, 3)


@pytest.mark.parametrize('ndim', [1, 3, 4])
def test_grid_unique_vs_oracle(dev, ndim):
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    rng = np.random.default_rng(ndim)
    hi = [3, 17, 40, 33][-ndim:]
    coors = np.stack([rng.integers(-1 if k == ndim - 1 else 0, hi[k], size=50_000) for k in range(ndim)], 1).astype(np.int32)
    outc, inv, counts = grid_unique(torch.from_numpy(coors).to(dev), dims=hi)
    eo, ei, ec = O.unique_rows(coors)
    assert np.array_equal(outc.cpu().numpy().reshape(len(eo), -1), eo)
    assert np.array_equal(inv.cpu().numpy(), ei)
    assert np.array_equal(counts.cpu().numpy(), ec)
    # bounds derived from the data (one sync) give the same answer
    outc2, inv2, _ = grid_unique(torch.from_numpy(coors).to(dev))
    assert torch.equal(inv, inv2) and torch.equal(outc, outc2)


@pytest.mark.parametrize('mode', ['max', 'mean', 'sum'])
@pytest.mark.parametrize('c', [3, 16, 128, 131])
def test_segment_reduce_fwd_bwd_vs_oracle(dev, mode, c):
    from objectcentricocccompletion_amd.voxel import segment_reduce
    rng = np.random.default_rng(c)
    n, segs = 20_011, 97
    inv = np.sort(rng.integers(0, segs, size=n)).astype(np.int32)  # grouped like pooled RoI points
    inv[rng.integers(0, n, size=50)] = -1                           # dropped rows
    if c == 16:
        inv = inv[rng.permutation(n)]                               # and a fully shuffled case
    feats = rng.standard_normal((n, c)).astype(np.float32)
    ft = torch.from_numpy(feats).to(dev).requires_grad_(True)
    out = segment_reduce(ft, torch.from_numpy(inv).to(dev), segs, mode)
    eo, ecnt, earg = O.segment_reduce(feats, inv, segs, mode)
    if mode == 'max':
        assert np.array_equal(out.detach().cpu().numpy(), eo)  # exact: integer atomics on bit patterns
    else:
        # float atomics add run partials in arrival order: fp32 rounding differs from the sequential oracle
        assert np.allclose(out.detach().cpu().numpy(), eo, rtol=1e-5, atol=5e-5)
    go = rng.standard_normal((segs, c)).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    exp = np.zeros_like(feats)
    valid = inv >= 0
    if mode == 'sum':
        exp[valid] = go[inv[valid]]
    elif mode == 'mean':
        exp[valid] = go[inv[valid]] / ecnt[inv[valid]][:, None]
    else:
        rows = np.arange(n)[:, None]
        hit = valid[:, None] & (earg[np.where(valid, inv, 0)] == rows)
        exp = np.where(hit, go[np.where(valid, inv, 0)], 0).astype(np.float32)
    assert np.allclose(ft.grad.cpu().numpy(), exp, rtol=1e-6, atol=1e-6)


def test_dynamic_scatter_reference_known_answer(dev):
    """The recipe of the reference's tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:8-93:
    200000x3 feats, coors in [-1,20)^3, brute-force expectation, allclose(atol=1e-2, rtol=1e-5),
    plus its empty-input and all-negative cases."""
    from objectcentricocccompletion_amd.voxel import DynamicScatter
    g = torch.Generator().manual_seed(0)
    feats = torch.rand(200000, 3, generator=g) * 100 - 50
    coors = torch.randint(-1, 20, (200000, 3), generator=g, dtype=torch.int32)
    dsmean = DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], True)
    dsmax = DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], False)
    e = dsmean(torch.zeros(0, 3, device=dev), torch.zeros(0, 3, dtype=torch.int32, device=dev))
    assert e[0].shape == (0, 3) and e[1].shape == (0, 3)
    neg = dsmax(feats[:10].to(dev), -torch.ones(10, 3, dtype=torch.int32, device=dev))
    assert neg[0].shape[0] == 0 and neg[1].shape[0] == 0
    ref_c = coors.unique(dim=0)
    ref_c = ref_c[ref_c.min(dim=-1).values >= 0]
    fm, cm = dsmean(feats.to(dev), coors.to(dev))
    fx, cx = dsmax(feats.to(dev), coors.to(dev))
    assert torch.equal(cm.cpu(), ref_c) and torch.equal(cx.cpu(), ref_c)
    eo, _, _, _ = O.dynamic_scatter(feats.numpy(), coors.numpy(), 'mean')
    assert np.allclose(fm.cpu().numpy(), eo, atol=1e-2, rtol=1e-5)
    for r in (0, 1234, len(ref_c) - 1):  # brute force, as the reference test does
        sel = feats[(coors == ref_c[r]).all(-1)]
        assert torch.allclose(fm[r].cpu(), sel.mean(0), atol=1e-2, rtol=1e-5)
        assert torch.equal(fx[r].cpu(), sel.max(0).values)


def test_scatter_properties_at_benchmark_scale(dev):
    """64 grids x 40^3, 2000 points each: sum of per-voxel sums == sum of points, counts add
    up, every point's voxel holds a max >= the point, second run bit-identical for max."""
    from objectcentricocccompletion_amd.voxel import dynamic_scatter, voxelization
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    g = torch.Generator().manual_seed(1)
    B, P = 64, 2000
    pts = ((torch.rand(B * P, 5, generator=g) * 2 - 1) * 4).to(dev)
    b = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(P)
    zyx = voxelization(pts, [0.2] * 3, [-4, -4, -4, 4, 4, 4], -1, -1)
    coors = torch.cat([b[:, None], zyx], 1)
    vc, inv, counts = grid_unique(coors, dims=[B, 40, 40, 40])
    assert int(counts.sum()) == B * P and vc.shape[0] == counts.shape[0]
    key = ((vc[:, 0].long() * 40 + vc[:, 1]) * 40 + vc[:, 2]) * 40 + vc[:, 3]
    assert bool((key[1:] > key[:-1]).all())            # sorted and unique
    assert torch.equal(vc[inv.long()], coors)          # inverse map is exact
    fsum, _ = dynamic_scatter(pts, coors, 'sum', grid_shape=[B, 40, 40, 40])
    assert torch.allclose(fsum.double().sum(0), pts.double().sum(0), rtol=1e-6, atol=1e-3)
    fmax, _ = dynamic_scatter(pts, coors, 'max', grid_shape=[B, 40, 40, 40])
    assert bool((fmax[inv.long()] >= pts).all())
    fmax2, _ = dynamic_scatter(pts, coors, 'max', grid_shape=[B, 40, 40, 40])
    assert torch.equal(fmax, fmax2)
