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
