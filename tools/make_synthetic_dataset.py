"""Write a small synthetic tracklet dataset in the reference's on-disk formats (what WaymoTrackletDatasetWithOcc
reads, mmdet3d/datasets/waymo_tracklet_dataset.py:491-584): <root>/tracklet_data/synth_training.pkl (proposals),
synth_training_gt_candidates.pkl, synth_training_database/<segment>--<id>.npy (per-frame [n,6] points),
<root>/poses.pkl, <root>/occ_gt/<segment>/<id>.npz (key 'occ', X x Y x Z in {0,1,2}).

usage: python tools/make_synthetic_dataset.py <root> [--tracklets 6] [--frames 40]"""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('root')
    ap.add_argument('--tracklets', type=int, default=6)
    ap.add_argument('--frames', type=int, default=40)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args(argv)
    rng = np.random.default_rng(a.seed)
    tdir = os.path.join(a.root, 'tracklet_data')
    db = os.path.join(tdir, 'synth_training_database')
    os.makedirs(db, exist_ok=True)
    proposals, candidates, poses = [], [], {}
    for t in range(a.tracklets):
        seg, tid = f'segment-{t // 3:03d}', f'obj{t:03d}'
        ts = [1_000_000 * (t + 1) + 100_000 * f for f in range(a.frames)]
        size = np.array([rng.uniform(1.7, 2.2), rng.uniform(4.0, 5.2), rng.uniform(1.4, 1.9)])
        boxes, pts = [], []
        for f, stamp in enumerate(ts):
            yaw = 0.02 * f + rng.normal(0, 0.01)
            pose = np.eye(4)
            pose[:3, :3] = rot_z(0.01 * f)
            pose[:3, 3] = [1.5 * f, 0.05 * f * f, 0]
            poses[stamp] = pose.astype(np.float32)
            ctr = np.array([12 + 0.3 * f, -4 + 0.1 * f, 0.1])           # in that frame's ego coordinates
            box = np.concatenate([ctr, size, [yaw]]).astype(np.float32)
            boxes.append(box)
            n = int(rng.integers(150, 400))
            local = (rng.random((n, 3)) - 0.5) * size * [1, 1, 1]         # points inside the box, box frame
            world = local @ rot_z(yaw).T + ctr + [0, 0, size[2] / 2]
            attr = rng.random((n, 3)).astype(np.float32)                  # intensity, elongation, (wrong) timestamp
            pts.append(np.concatenate([world, attr], 1).astype(np.float32))
        gt = [b + rng.normal(0, [0.05, 0.05, 0.02, 0.02, 0.02, 0.02, 0.01]).astype(np.float32) for b in boxes]
        far = [b + np.array([25, 25, 0, 0, 0, 0, 0], np.float32) for b in boxes]
        num_pts = [len(p) for p in pts]
        proposals.append((seg, tid, 1, False, [b[None] for b in boxes], ts, rng.uniform(0.3, 1.0, a.frames).tolist(), num_pts))
        candidates.append([(seg, tid + '_far', 1, False, [b[None] for b in far], ts, [1.0] * a.frames, num_pts),
                           (seg, tid + '_gt', 1, False, [b[None] for b in gt], ts, [1.0] * a.frames, num_pts)])
        arr = np.empty(len(pts), dtype=object)
        for i, p in enumerate(pts):
            arr[i] = p
        np.save(os.path.join(db, f'{seg}--{tid}.npy'), arr, allow_pickle=True)
        for name in (tid + '_gt', tid + '_far'):
            dims = np.ceil(size / 0.2).astype(int)
            occ = rng.integers(0, 3, dims).astype(np.int64)
            os.makedirs(os.path.join(a.root, 'occ_gt', seg), exist_ok=True)
            np.savez_compressed(os.path.join(a.root, 'occ_gt', seg, f'{name}.npz'), occ=occ)
    with open(os.path.join(tdir, 'synth_training.pkl'), 'wb') as f:
        pickle.dump(proposals, f)
    with open(os.path.join(tdir, 'synth_training_gt_candidates.pkl'), 'wb') as f:
        pickle.dump(candidates, f)
    with open(os.path.join(a.root, 'poses.pkl'), 'wb') as f:
        pickle.dump(poses, f)
    print('wrote', a.tracklets, 'tracklets x', a.frames, 'frames under', a.root)


if __name__ == '__main__':
    main()
