"""the visibility ray test of the GT-occupancy annotation (ococc_occ_visibility_f64) against
vectors produced by the reference's own point_cloud_to_range_image_idx (tools/occ/occ_annotate.py:141-207,
executed from the reference file by oracle/gen_golden_annotate.py) and the labels annotate_trk derives (:519-556)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'occ_annotate.npz'))


def _ego(gold, k):
    from objectcentricocccompletion_amd.occ.annotate import frame_affines_from_boxes
    a = frame_affines_from_boxes(gold['boxes'])[k]
    return gold['centers'] @ a[:9].reshape(3, 3).T + a[9:]


def test_range_image_index_matches_reference(dev, gold):
    from objectcentricocccompletion_amd.occ.annotate import point_cloud_to_range_image_idx
    S, F = gold['extrinsics'].shape[:2]
    H, W = [int(v) for v in gold['size']]
    pts = torch.from_numpy(np.stack([_ego(gold, k) for k in range(F)], 0)).to(dev)
    for s in range(S):
        idx, rng = point_cloud_to_range_image_idx(pts, torch.from_numpy(gold['extrinsics'][s]),
                                                  torch.from_numpy(gold['inclinations'][s]), (H, W))
        assert idx.dtype == torch.int32 and rng.dtype == torch.float64
        assert np.allclose(rng.cpu().numpy(), gold[f'range_{s}'], rtol=1e-12, atol=1e-10)
        same = idx.cpu().numpy() == gold[f'idx_{s}']
        # integer pixel indices: identical except where atan2 lands within rounding of a pixel / beam boundary
        assert same.all(-1).mean() > 0.9995, same.all(-1).mean()
        assert (np.abs(idx.cpu().numpy()[..., 0] - gold[f'idx_{s}'][..., 0]) <= 1).all()


def test_visibility_labels_match_reference(dev, gold):
    from objectcentricocccompletion_amd.occ.annotate import visibility_ray_test
    S, F = gold['extrinsics'].shape[:2]
    imgs = [[torch.from_numpy(gold[f'range_image_{s}'][k]) for k in range(F)] for s in range(S)]
    vis = visibility_ray_test(torch.from_numpy(gold['centers']).to(dev), gold['boxes'], gold['extrinsics'],
                              gold['inclinations'], imgs)
    exp = gold['visibility']
    assert vis.dtype == torch.int32 and set(np.unique(vis.cpu().numpy())) <= {0, 2}
    assert 0.2 < (exp == 2).mean() < 0.8                       # the fixture has both labels in numbers
    assert (vis.cpu().numpy() == exp).mean() > 0.999
    # float64 range images take the same path
    imgs64 = [[i.double() for i in row] for row in imgs]
    vis64 = visibility_ray_test(torch.from_numpy(gold['centers']).to(dev), gold['boxes'], gold['extrinsics'],
                                gold['inclinations'], imgs64)
    assert torch.equal(vis, vis64)


def test_visibility_empty_and_never_seen(dev, gold):
    from objectcentricocccompletion_amd.occ.annotate import visibility_ray_test
    S, F = gold['extrinsics'].shape[:2]
    H, W = [int(v) for v in gold['size']]
    near = [[torch.full((H, W), 0.5) for _ in range(F)] for _ in range(S)]      # every return in front of the object
    far = [[torch.full((H, W), 500.0) for _ in range(F)] for _ in range(S)]     # every ray passes through
    c = torch.from_numpy(gold['centers']).to(dev)
    assert int(visibility_ray_test(c, gold['boxes'], gold['extrinsics'], gold['inclinations'], near).sum()) == 0
    assert bool((visibility_ray_test(c, gold['boxes'], gold['extrinsics'], gold['inclinations'], far) == 2).all())
    assert visibility_ray_test(c[:0], gold['boxes'], gold['extrinsics'], gold['inclinations'], far).numel() == 0
