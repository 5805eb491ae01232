"""Generate tests/golden/roi_corners.npz: LiDARInstance3DBoxes.corners of the REFERENCE (lidar_box3d.py:54-92) on seeded
boxes and the per-point RoI corner offsets TrackletRoIHeadOCC._bbox_forward appends under with_roi_corners
(tracklet_roi_head_occ.py:861-868, evaluated with the reference's own box class).  Imported through oracle/ref_shim.py in
the build container only; data only, no reference source."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.gen_golden_tta import load_tracklet_classes, make_boxes  # noqa: E402


def main():
    Boxes, _ = load_tracklet_classes()
    g = torch.Generator().manual_seed(21)
    n, p = 9, 40
    boxes = make_boxes(g, n)
    corners = Boxes(boxes).corners                                   # [n, 8, 3]
    rois = torch.cat([torch.randint(0, 3, (n, 1), generator=g).float(), boxes], 1)   # (batch, x, y, z, dx, dy, dz, yaw)
    roi_inds = torch.randint(0, n, (p,), generator=g)
    xyz = torch.randn(p, 3, generator=g) * 5
    # the statements of tracklet_roi_head_occ.py:862-868 on these inputs ("centers" are the first three roi columns there)
    c = Boxes(rois[:, 1:]).corners.to(xyz.dtype)
    c = torch.cat([c, rois[:, :3][:, None, :]], 1)
    offsets = (c[roi_inds] - xyz[:, None, :]).reshape(p, 27) / 10
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'roi_corners.npz')
    np.savez_compressed(path, boxes=boxes.numpy(), corners=corners.numpy(), rois=rois.numpy(), roi_inds=roi_inds.numpy(),
                        xyz=xyz.numpy(), offsets=offsets.numpy())
    print('wrote', path)


if __name__ == '__main__':
    main()
