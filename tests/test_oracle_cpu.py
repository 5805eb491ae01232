"""CPU suite, part 1: the oracle against the reference's golden vectors and against
independent formulations.  (No GPU, no /root/reference needed.)"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O


@pytest.fixture(scope='module')
def vox(golden_dir):
    return np.load(os.path.join(golden_dir, 'voxelize.npz'))


@pytest.mark.parametrize('name', ['grid40', 'box', 'kitti'])
def test_dynamic_voxelize_matches_reference_cpp(vox, name):
    # expected values come from the reference's own voxelization_cpu.cpp (oracle/gen_golden_voxel.py)
    c = O.dynamic_voxelize(vox[name + '_points'], vox[name + '_voxel_size'], vox[name + '_range'])
    assert np.array_equal(c, vox[name + '_dyn_coors'])


@pytest.mark.parametrize('name', ['grid40', 'box', 'kitti'])
@pytest.mark.parametrize('caps', [(5, 300), (35, 20000), (1, 7)])
def test_hard_voxelize_matches_reference_cpp(vox, name, caps):
    mp, mv = caps
    v, c, n = O.hard_voxelize(vox[name + '_points'], vox[name + '_voxel_size'], vox[name + '_range'], mp, mv)
    key = f'{name}_hard_{mp}_{mv}'
    assert np.array_equal(v, vox[key + '_voxels'])
    assert np.array_equal(c, vox[key + '_coors'])
    assert np.array_equal(n, vox[key + '_npv'])


def _random_voxels(rng, B, shape, density, shuffle=True):
    mask = rng.random((B,) + tuple(shape)) < density
    idx = np.argwhere(mask).astype(np.int32)
    if shuffle:
        idx = idx[rng.permutation(len(idx))]
    return idx


def test_rulebook_offset_convention_kat():
    # observed on the compiled reference during the survey:
    # input (1,1,1) -> output (2,2,2) is kernel offset 0; the centre is offset 13
    pairs, num = O.subm_rulebook(np.array([[0, 1, 1, 1], [0, 2, 2, 2]], np.int32), 1, (4, 4, 4))
    assert num.tolist() == [1] + [0] * 12 + [2] + [0] * 12 + [1]
    assert pairs[0, :, 0].tolist() == [0, 1]
    assert pairs[26, :, 0].tolist() == [1, 0]
    assert pairs[13, :, :2].tolist() == [[0, 1], [0, 1]]


def test_subm_conv_equals_dense_conv3d():
    rng = np.random.default_rng(1)
    B, shape = 2, (6, 7, 8)
    idx = _random_voxels(rng, B, shape, 0.3)
    N, cin, cout = len(idx), 5, 7
    x = rng.standard_normal((N, cin)).astype(np.float32)
    w = rng.standard_normal((3, 3, 3, cin, cout)).astype(np.float32)
    dy = rng.standard_normal((N, cout)).astype(np.float32)
    pairs, num = O.subm_rulebook(idx, B, shape)
    assert num[13] == N and num.sum() == (pairs[:, 0] >= 0).sum()
    y = O.indice_conv(x, w, pairs, num, N, subm=True)
    din, dw = O.indice_conv_backward(x, w, dy, pairs, num, subm=True)

    I = [torch.from_numpy(idx[:, i]).long() for i in range(4)]
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    dense = torch.zeros((B,) + shape + (cin,)).index_put((I[0], I[1], I[2], I[3]), xt)
    yd = torch.nn.functional.conv3d(dense.permute(0, 4, 1, 2, 3), wt.permute(4, 3, 0, 1, 2), padding=1)
    ysel = yd.permute(0, 2, 3, 4, 1)[I[0], I[1], I[2], I[3]]
    ysel.backward(torch.from_numpy(dy))
    assert np.allclose(y, ysel.detach().numpy(), atol=1e-4)
    assert np.allclose(din, xt.grad.numpy(), atol=1e-4)
    assert np.allclose(dw, wt.grad.numpy(), atol=1e-4)


def test_strided_conv_equals_dense_conv3d():
    rng = np.random.default_rng(2)
    B, shape = 2, (6, 7, 8)
    idx = _random_voxels(rng, B, shape, 0.3)
    N, cin, cout = len(idx), 4, 6
    x = rng.standard_normal((N, cin)).astype(np.float32)
    w = rng.standard_normal((3, 3, 3, cin, cout)).astype(np.float32)
    oshape = [(s + 2 - 3) // 2 + 1 for s in shape]
    outi, pairs, num = O.conv_rulebook(idx, B, oshape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    y = O.indice_conv(x, w, pairs, num, len(outi))
    dense = torch.zeros((B,) + shape + (cin,))
    dense[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = torch.from_numpy(x)
    yd = torch.nn.functional.conv3d(dense.permute(0, 4, 1, 2, 3), torch.from_numpy(w).permute(4, 3, 0, 1, 2),
                                    padding=1, stride=2).permute(0, 2, 3, 4, 1)
    assert np.allclose(y, yd[outi[:, 0], outi[:, 1], outi[:, 2], outi[:, 3]].numpy(), atol=1e-4)
    # every non-empty dense output site is an active output of the rulebook
    assert int((yd.abs().sum(-1) > 0).sum()) <= len(outi)


def test_unique_and_segment_reduce_match_torch():
    rng = np.random.default_rng(3)
    coors = rng.integers(-1, 20, size=(5000, 3)).astype(np.int32)
    feats = rng.standard_normal((5000, 6)).astype(np.float32)
    outc, inv, counts = O.unique_rows(coors)
    keep = (coors >= 0).all(1)
    tc, tinv, tcnt = torch.unique(torch.from_numpy(coors[keep]), dim=0, return_inverse=True, return_counts=True)
    assert np.array_equal(outc, tc.numpy()) and np.array_equal(inv[keep], tinv.numpy())
    assert np.array_equal(counts, tcnt.numpy()) and (inv[~keep] == -1).all()
    for mode, red in (('max', 'amax'), ('mean', 'mean'), ('sum', 'sum')):
        out, cnt, arg = O.segment_reduce(feats, inv, len(outc), mode)
        ref = torch.zeros(len(outc), 6).scatter_reduce(0, tinv[:, None].expand(-1, 6), torch.from_numpy(feats[keep]),
                                                       red, include_self=False)
        assert np.allclose(out, ref.numpy(), atol=1e-5)
        if mode == 'max':
            rows = np.arange(5000)
            for g in (0, 7, len(outc) - 1):
                for ch in range(6):
                    cand = rows[(inv == g) & (feats[:, ch] == out[g, ch])]
                    assert arg[g, ch] == cand.min()


def test_dynamic_scatter_recipe_of_reference_test():
    # tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:56-65 (brute force expectation)
    rng = np.random.default_rng(4)
    feats = rng.random((3000, 3)).astype(np.float32) * 100 - 50
    coors = rng.integers(-1, 8, size=(3000, 3)).astype(np.int32)
    for mode in ('mean', 'max'):
        out, outc, inv, _ = O.dynamic_scatter(feats, coors, mode)
        ref_c = np.unique(coors[(coors >= 0).all(1)], axis=0)
        assert np.array_equal(outc, ref_c)
        for r in (0, len(ref_c) // 2, len(ref_c) - 1):
            sel = feats[(coors == ref_c[r]).all(1)]
            exp = sel.mean(0) if mode == 'mean' else sel.max(0)
            assert np.allclose(out[r], exp, atol=1e-4)


def test_bf16_round_matches_torch():
    a = np.random.default_rng(5).standard_normal(4096).astype(np.float32) * 100
    assert np.array_equal(O.bf16_round(a), torch.from_numpy(a).to(torch.bfloat16).float().numpy())
