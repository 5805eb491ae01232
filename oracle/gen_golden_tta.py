"""Generate tests/golden/tta.npz: the REFERENCE's test-time-augmentation bookkeeping on seeded tracklets --
LiDARTracklet.flip / rotate / scale / translate (mmdet3d/core/bbox/structures/lidar_tracklet.py:253-276 over
LiDARInstance3DBoxes, lidar_box3d.py:143-216, base_box3d.py:156-231) and LiDARTracklet.merge_augs (:552-607)
in the modes that need no compiled extension ('max', 'weighted').  Imported through oracle/ref_shim.py in
the build container only; data only, no reference source."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_shim as R  # noqa: E402


def load_tracklet_classes():
    R.install()
    # names the box classes import at module level and never touch on the paths exercised here
    sys.modules['mmdet3d.ops.iou3d'].iou3d_cuda = None
    pts = sys.modules.get('mmdet3d.core.points') or types.ModuleType('mmdet3d.core.points')
    pts.BasePoints = type('BasePoints', (), {})
    sys.modules['mmdet3d.core.points'] = pts
    sys.modules['mmdet3d.ops.roiaware_pool3d'].points_in_boxes_gpu = None
    R.load('mmdet3d.core.bbox.structures.base_box3d')
    box = R.load('mmdet3d.core.bbox.structures.lidar_box3d')
    st = sys.modules['mmdet3d.core.bbox.structures']
    st.LiDARInstance3DBoxes = box.LiDARInstance3DBoxes
    trk = R.load('mmdet3d.core.bbox.structures.lidar_tracklet')
    return box.LiDARInstance3DBoxes, trk.LiDARTracklet


def make_boxes(g, n):
    b = torch.zeros(n, 7)
    b[:, :3] = torch.randn(n, 3, generator=g) * torch.tensor([30., 30., 1.])
    b[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([1.0, 3.0, 0.8]) + torch.tensor([1.6, 3.5, 1.3])
    b[:, 6] = (torch.rand(n, generator=g) * 2 - 1) * 3.1
    return b


def main():
    Boxes, Trk = load_tracklet_classes()
    g = torch.Generator().manual_seed(21)
    out = {}
    L = 9
    base = make_boxes(g, L)
    out['boxes'] = base.numpy().copy()

    def tracklet(b, scores=None):
        # merge_augs expects numpy boxes [1,7] and float scores; the transforms expect box objects
        t = Trk('seg', 'id0', 1, False, box_list=[Boxes(b[i:i + 1].clone()) for i in range(len(b))],
                ts_list=list(range(len(b))), score_list=list(scores) if scores is not None else [1.0] * len(b))
        return t

    def cat(t):
        return torch.cat([x.tensor for x in t.box_list], 0).numpy().copy()

    t = tracklet(base); t.flip('horizontal'); out['flip_h'] = cat(t)
    t = tracklet(base); t.flip('vertical'); out['flip_v'] = cat(t)
    t = tracklet(base); t.rotate(0.37); out['rot'] = cat(t)
    t = tracklet(base); t.rotate(-1.9); out['rot_neg'] = cat(t)
    t = tracklet(base); t.scale(1.07); out['scale'] = cat(t)
    t = tracklet(base); t.translate(torch.tensor([0.5, -1.25, 0.2])); out['translate'] = cat(t)
    t = tracklet(base); t.flip('horizontal'); t.rotate(0.37); t.rotate(-0.37); t.flip('horizontal'); out['round_trip'] = cat(t)

    for num_augs in (3, 4):
        aug_boxes = torch.stack([base + torch.randn(L, 7, generator=g) * 0.05 for _ in range(num_augs)], 0)
        aug_scores = torch.rand(num_augs, L, generator=g) * 0.9 + 0.05
        out[f'aug_boxes_{num_augs}'] = aug_boxes.numpy().copy()
        out[f'aug_scores_{num_augs}'] = aug_scores.numpy().copy()
        for mode in ('max', 'weighted'):
            res = []
            for a in range(num_augs):
                tr = Trk('seg', 'id0', 1, False, box_list=None)
                tr.box_list = [aug_boxes[a, i:i + 1].numpy().astype(np.float64) for i in range(L)]
                tr.score_list = [float(s) for s in aug_scores[a]]
                tr.ts_list = list(range(L))
                res.append(tr)
            m = Trk.merge_augs(res, dict(merge=mode))
            out[f'merge_{mode}_{num_augs}_boxes'] = np.concatenate(m.box_list, 0)
            out[f'merge_{mode}_{num_augs}_scores'] = np.asarray(m.score_list)
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'tta.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
