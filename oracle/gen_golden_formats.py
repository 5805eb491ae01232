"""Generate tests/golden/formats.npz: the on-disk tracklet tuple of the REFERENCE's own LiDARTracklet
(mmdet3d/core/bbox/structures/lidar_tracklet.py:29-128 construction / append / freeze, :130-138 to_dump_format,
:157-161 from_dump_format) -- what the *_training.pkl / *_gt_candidates.pkl files of WaymoTrackletDataset hold
(waymo_tracklet_dataset.py:83-106).  The tracklet is built the way tools/ctrl/utils.py:18-58 builds GT tracklets (frames
appended in time order, frozen: freeze() asserts sorted timestamps), dumped, pickled and reloaded by the reference class; the arrays of the tuple and of
the reloaded object go into the .npz (data only).  Imported through oracle/ref_shim.py in the build container only."""
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.gen_golden_tta import load_tracklet_classes, make_boxes  # noqa: E402


def main():
    Boxes, Trk = load_tracklet_classes()
    g = torch.Generator().manual_seed(33)
    L = 7
    base = make_boxes(g, L)
    ts = [1553000000000000 + 100000 * i for i in range(L)]
    scores = torch.rand(L, generator=g).tolist()
    order = list(range(L))
    t = Trk('segment-1234_with_camera_labels', 'AbCdEf-123', 1, False)
    for i in order:
        t.append(Boxes(base[i:i + 1].clone(), box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)), scores[i], ts[i], False)
    t.freeze()
    t.num_pts_in_boxes = [5 * i + 1 for i in range(L)]
    item = t.to_dump_format()
    blob = pickle.dumps(item, protocol=2)
    back = Trk.from_dump_format(pickle.loads(blob))
    out = dict(
        segment_name=np.array(item[0]), id=np.array(item[1]), type=np.int64(item[2]), in_world=np.bool_(item[3]),
        boxes=np.concatenate(item[4], 0), box_shape=np.array(item[4][0].shape), ts=np.array(item[5], dtype=np.int64),
        scores=np.array(item[6], dtype=np.float64), num_pts=np.array(item[7], dtype=np.int64),
        tuple_len=np.int64(len(item)),
        # the reloaded object: what the dataset works with
        back_len=np.int64(len(back)), back_ts=np.array(back.ts_list, dtype=np.int64),
        back_boxes=np.concatenate([b if isinstance(b, np.ndarray) else b.tensor.numpy() for b in back.box_list], 0),
        back_size=np.int64(back.size), appended_boxes=base.numpy(), appended_order=np.array(order))
    dst = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'formats.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    main()
