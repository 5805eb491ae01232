"""The single-launch forms of the target glue (csrc/target_ops.hip) against the torch expressions they replace --
rotation_3d_in_axis (mmdet3d/core/bbox/structures/utils.py:21-61), the GT-frame -> RoI-frame chain of the occupancy
samples (ococc_bbox_head.py:1279-1290) and the canonical box targets + DeltaXYZWLHRBBoxCoder.encode
(ococc_bbox_head.py:1190-1222, delta_xyzwhlr_bbox_coder.py:21-50) -- on the same device inputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device('cuda:0')


def _boxes(n, g, dev, wide=False):
    b = torch.randn(n, 9 if wide else 7, generator=g)
    b[:, 3:6] = b[:, 3:6].abs() + 0.5
    b[:, 6] = (torch.rand(n, generator=g) - 0.5) * 4 * math.pi      # yaws beyond one turn, both signs
    return b.to(dev)


def _slow(monkeypatch):
    from objectcentricocccompletion_amd import bbox, heads
    monkeypatch.setattr(bbox, '_plain', lambda *t: False)
    monkeypatch.setattr(heads, '_plain', lambda *t: False)


@pytest.mark.parametrize('n,m', [(1, 5000), (37, 512), (128, 1), (0, 4)])
def test_rotation_about_z_in_one_launch(dev, monkeypatch, n, m):
    from objectcentricocccompletion_amd.bbox import rotation_3d_in_axis
    g = torch.Generator().manual_seed(0)
    p, a = torch.randn(n, m, 3, generator=g).to(dev) * 10, ((torch.rand(n, generator=g) - 0.5) * 7).to(dev)
    fast = rotation_3d_in_axis(p, a, axis=2)
    _slow(monkeypatch)
    ref = rotation_3d_in_axis(p, a, axis=2)
    assert fast.shape == ref.shape
    if n:
        assert float((fast - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # a differentiated input keeps the torch expression (and its gradient)
    monkeypatch.undo()
    q = p.clone().requires_grad_(True)
    out = rotation_3d_in_axis(q, a, axis=2)
    assert out.requires_grad


@pytest.mark.parametrize('n,m', [(40, 512), (3, 1), (0, 512)])
def test_points_from_gt_frame_to_roi_frame_in_one_launch(dev, monkeypatch, n, m):
    from objectcentricocccompletion_amd.bbox import points_box_to_box
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(n, m, 3, generator=g).to(dev) * 3
    gt, roi = _boxes(n, g, dev, wide=True)[:, :7], _boxes(n, g, dev)      # (a column slice of a wider tensor: read in place)
    fast = points_box_to_box(xyz.clone(), gt, roi)
    _slow(monkeypatch)
    ref = points_box_to_box(xyz.clone(), gt, roi)
    assert fast.shape == ref.shape
    if n:
        assert float((fast - ref).abs().max()) <= 2e-6 * float(ref.abs().max().clamp(min=1.0))


@pytest.mark.parametrize('n', [300, 1, 0])
def test_canonical_box_targets_in_one_launch(dev, monkeypatch, n):
    from objectcentricocccompletion_amd.heads import OccBBoxHead
    from objectcentricocccompletion_amd.bbox import DeltaXYZWLHRBBoxCoder

    class Stub(object):   # (the method reads only the coder)
        bbox_coder = DeltaXYZWLHRBBoxCoder()
    g = torch.Generator().manual_seed(2)
    roi, gt = _boxes(n, g, dev), _boxes(n, g, dev)
    if n > 10:   # yaw differences on the branch boundaries of the canonical angle
        gt[:4, 6] = roi[:4, 6] + torch.tensor([math.pi / 2, 1.5 * math.pi, math.pi, 0.0], device=dev)
    fast = OccBBoxHead._canonical_box_targets(Stub(), roi, gt)
    _slow(monkeypatch)
    ref = OccBBoxHead._canonical_box_targets(Stub(), roi, gt)
    assert fast.shape == ref.shape == (n, 7)
    if n:
        err = (fast - ref).abs()
        # (the angle column may sit on either side of a branch where the difference is within rounding of pi / 2 or 3 pi / 2:
        # the clamp to [-pi / 2, pi / 2] then gives +-pi / 2 -- compared modulo pi there)
        ang = err[:, 6]
        ang = torch.minimum(ang, (ang - math.pi).abs())
        assert float(err[:, :6].max()) <= 1e-5 * float(ref[:, :6].abs().max().clamp(min=1.0)) and float(ang.max()) <= 1e-5
