"""Synthetic Waymo-shaped tracklet batches for benchmarks, smoke runs and the
training script until the real tracklet dataset lands: per sample one vehicle tracklet of L
frames, points decorated as PointDecoration does (mmdet3d/datasets/pipelines/
tracklet_pipelines.py:582-607: intensity, elongation, yaw/pi, size/10 x3, score), one GT
candidate tracklet, K occupancy query points (xyz in the GT box frame + label 0/1/2) with a
label confidence."""
import math

import torch

from .tracklet import Tracklet


def synthetic_training_batch(num_tracklets, frames=32, pts_per_frame=64, occ_queries=512, seed=0,
                             device='cuda'):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g)
    points, pts_frames, trks, cands, occs, occ_scores = [], [], [], [], [], []
    for b in range(num_tracklets):
        w, l, h = 1.8 + 0.4 * rnd(1), 4.2 + 0.8 * rnd(1), 1.5 + 0.4 * rnd(1)
        x0, y0, z0 = rnd(1) * 100 - 50, rnd(1) * 100 - 50, -1.5 + rnd(1)
        heading = (rnd(1) * 2 - 1) * math.pi
        t = torch.arange(frames, dtype=torch.float32)
        yaw = heading + 0.02 * torch.randn(frames, generator=g)
        boxes = torch.stack([x0 + t * torch.cos(heading), y0 + t * torch.sin(heading), z0.expand(frames),
                             w.expand(frames), l.expand(frames), h.expand(frames), yaw], 1)
        score = 0.3 + 0.7 * rnd(frames)
        n = frames * pts_per_frame
        fr = torch.arange(frames).repeat_interleave(pts_per_frame)
        loc = (rnd(n, 3) - 0.5) * (torch.cat([l, w, h]) + 0.5)          # box frame, x along l
        c, s = torch.cos(boxes[fr, 6]), torch.sin(boxes[fr, 6])
        xyz = torch.stack([boxes[fr, 0] + loc[:, 0] * c - loc[:, 1] * s,
                           boxes[fr, 1] + loc[:, 0] * s + loc[:, 1] * c,
                           boxes[fr, 2] + boxes[fr, 5] / 2 + loc[:, 2]], 1)
        deco = torch.cat([rnd(n, 2), boxes[fr, 6:7] / math.pi, boxes[fr, 3:6] / 10, score[fr][:, None]], 1)
        perm = torch.randperm(n, generator=g)
        points.append(torch.cat([xyz, deco], 1)[perm].to(device))
        pts_frames.append(fr[perm].to(device))
        ts = list(range(10000 * b, 10000 * b + frames))
        trks.append(Tracklet(boxes.to(device), ts, score.to(device), type=0))
        gt = boxes + torch.randn(frames, 7, generator=g) * torch.tensor([0.1, 0.1, 0.05, 0.05, 0.05, 0.05, 0.02])
        cands.append([Tracklet(gt.to(device), ts, type=0)])
        q = (rnd(occ_queries, 3) - 0.5) * torch.cat([l, w, h])
        lab = torch.randint(0, 3, (occ_queries, 1), generator=g).float()
        occs.append([torch.cat([q, lab], 1).to(device)])
        occ_scores.append([torch.tensor([0.9], device=device)])
    return dict(points=points, pts_frame_inds=pts_frames, img_metas=None, tracklet=trks,
                gt_tracklet_candidates=cands, occ_labels=occs, occ_labels_scores=occ_scores)
