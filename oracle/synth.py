"""Deterministic synthetic weights and tracklet inputs shared by the golden generators (build
container, reference side) and the tests (GPU box, product side).  Weights are a function of
the parameter NAME and SHAPE only, so loading them by name into either implementation also
checks state-dict compatibility.  TEST INFRASTRUCTURE ONLY."""
import hashlib
import math

import numpy as np
import torch


def synth_tensor(name, shape, seed=0):
    h = int.from_bytes(hashlib.sha256(f'{seed}:{name}'.encode()).digest()[:8], 'little') % (2 ** 31)
    g = torch.Generator().manual_seed(h)
    shape = tuple(shape)
    if len(shape) >= 2:  # Linear / in_proj weights: variance preserving
        fan_in = shape[-1]
        return torch.randn(shape, generator=g) / math.sqrt(fan_in)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'weight':  # LayerNorm gamma
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)  # biases / LN beta


def synth_state_dict(shapes, seed=0):
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def synth_tracklets(num_tracklets=2, frames=32, pts_per_frame=60, seed=0, first_frame=0):
    """Waymo-shaped vehicle tracklets: boxes (x,y,z_bottom,w,l,h,yaw) on a smooth
    path, points uniform in the box enlarged by 0.25 m (some outside the 0.5 m margin too),
    2 attributes; returns numpy arrays ready for the pooling op."""
    rng = np.random.default_rng(seed)
    rois, roi_frames, pts, pts_batch, pts_frame, attrs = [], [], [], [], [], []
    for b in range(num_tracklets):
        w, l, h = rng.uniform(1.8, 2.2), rng.uniform(4.2, 5.0), rng.uniform(1.5, 1.9)
        x0, y0, z0 = rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-1.5, -0.5)
        heading = rng.uniform(-np.pi, np.pi)
        for t in range(frames):
            cx, cy = x0 + t * np.cos(heading), y0 + t * np.sin(heading)
            yaw = heading + rng.normal(0, 0.02)
            rois.append([b, cx, cy, z0, w, l, h, yaw])
            roi_frames.append(first_frame + t)
            n = pts_per_frame if (b + t) % 7 else 0  # a few empty RoIs
            if n == 0:
                continue
            loc = (rng.random((n, 3)) - 0.5) * (np.array([l, w, h]) + 0.9)  # box frame, x along l
            c, s = np.cos(yaw), np.sin(yaw)  # inverse of mmdet3d's lidar_to_local_coords
            gx = cx + loc[:, 0] * c - loc[:, 1] * s
            gy = cy + loc[:, 0] * s + loc[:, 1] * c
            gz = z0 + h / 2 + loc[:, 2]
            pts.append(np.stack([gx, gy, gz], 1))
            pts_batch.append(np.full(n, b))
            pts_frame.append(np.full(n, first_frame + t))
            attrs.append(rng.random((n, 2)))
    rois = np.asarray(rois, np.float32)
    pts = np.concatenate(pts).astype(np.float32)
    perm = rng.permutation(len(pts))  # points arrive in no particular order
    return dict(rois=rois, roi_frame_inds=np.asarray(roi_frames, np.int64), pts_xyz=pts[perm],
                pts_batch=np.concatenate(pts_batch).astype(np.int64)[perm],
                pts_frame=np.concatenate(pts_frame).astype(np.int64)[perm],
                pts_attr=np.concatenate(attrs).astype(np.float32)[perm])


def synth_training_scene(seed=0, frames=16, pts_per_frame=48, occ_queries=64):
    """Four proposal tracklets with their GT candidates, decorated points and occupancy labels -- the inputs of
    TrackletRoIHeadOCC.forward_train / simple_test -- as plain numpy, so that the golden generator can wrap them
    in the reference's LiDARTracklet and the tests in the product's Tracklet.  Covers: two candidates (far /
    near), a candidate missing timestamps (negative RoIs), a frame without points (empty RoI), an occupancy
    label score under occ_label_thresh, and a tracklet without any candidate."""
    rng = np.random.default_rng(seed)
    samples = []
    for b in range(4):
        w, l, h = rng.uniform(1.8, 2.2), rng.uniform(4.2, 5.0), rng.uniform(1.5, 1.9)
        x0, y0, z0 = rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-1.5, -0.5)
        heading = rng.uniform(-np.pi, np.pi)
        t = np.arange(frames)
        boxes = np.stack([x0 + t * np.cos(heading), y0 + t * np.sin(heading), np.full(frames, z0), np.full(frames, w),
                          np.full(frames, l), np.full(frames, h), heading + rng.normal(0, 0.02, frames)], 1).astype(np.float32)
        scores = rng.uniform(0.3, 1.0, frames).astype(np.float32)
        ts = [1500000000000000 + 100000 * (1000 * b + 7 * i) for i in range(frames)]  # Waymo microsecond stamps (> 1e10, lidar_tracklet.py:180)
        pts, fr = [], []
        for i in range(frames):
            if b == 1 and i == 3:
                continue  # a frame without points: an empty RoI that is assigned a GT box
            loc = (rng.random((pts_per_frame, 3)) - 0.5) * (np.array([l, w, h]) + 0.9)
            yaw = boxes[i, 6]
            c, s = np.cos(yaw), np.sin(yaw)
            xyz = np.stack([boxes[i, 0] + loc[:, 0] * c - loc[:, 1] * s, boxes[i, 1] + loc[:, 0] * s + loc[:, 1] * c,
                            boxes[i, 2] + h / 2 + loc[:, 2]], 1)
            deco = np.concatenate([rng.random((pts_per_frame, 2)), np.full((pts_per_frame, 1), yaw / np.pi),
                                   np.tile(boxes[i, 3:6] / 10, (pts_per_frame, 1)), np.full((pts_per_frame, 1), scores[i])], 1)
            pts.append(np.concatenate([xyz, deco], 1))
            fr.append(np.full(pts_per_frame, i))
        pts, fr = np.concatenate(pts).astype(np.float32), np.concatenate(fr).astype(np.int64)
        perm = rng.permutation(len(pts))
        noise = np.array([0.1, 0.1, 0.05, 0.05, 0.05, 0.05, 0.02]) * (3.0 if b == 2 else 1.0)
        near = (boxes + rng.normal(0, 1, boxes.shape) * noise).astype(np.float32)
        far = near.copy()
        far[:, :2] += 1.5
        occ = lambda: np.concatenate([(rng.random((occ_queries, 3)) - 0.5) * [l, w, h],
                                      rng.integers(0, 3, (occ_queries, 1))], 1).astype(np.float32)
        if b == 0:
            cands = [(far, ts, occ(), 0.9), (near, ts, occ(), 0.8)]
        elif b == 1:
            keep = [i for i in range(frames) if i % 5 != 2]  # candidate missing some timestamps
            cands = [(near[keep], [ts[i] for i in keep], occ(), 0.95)]
        elif b == 2:
            cands = [(near, ts, occ(), 0.3)]  # label score under occ_label_thresh = 0.4
        else:
            cands = []
        samples.append(dict(points=pts[perm], pts_frame_inds=fr[perm], boxes=boxes, scores=scores, ts=ts,
                            candidates=cands))
    return samples
