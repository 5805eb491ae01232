"""Tracklet training / test pipeline transforms (the in-memory part): the steps of
configs/ococc/ococcnet.py's train_pipeline that act on points + tracklets, with the reference's names,
constructor arguments, result-dict keys and random-number call order
(mmdet3d/datasets/pipelines/tracklet_pipelines.py:175-225 TrackletRegularization, :306-465
TrackletGlobalRotScaleTrans, :467-553 TrackletRandomFlip, :555-623 PointDecoration, :626-651 FrameDropout,
:654-678 TrackletNoise; LiDARTracklet.add_*_noise lidar_tracklet.py:500-551).

Containers: results['points'] is a list of per-frame [n_i, C] tensors (or, after a concatenating step, one
[N, C] tensor), results['pts_frame_inds'] the matching frame-index tensors, results['tracklet'] a
tracklet.Tracklet (boxes [L,7] tensor), results['gt_tracklet_candidates'] a list of them.  File loading
(LoadTrackletPoints, LoadTrackletAnnotations, LoadAnnotationsOcc), the pose transform and the dataset class are
not part of this module (they are bound to the Waymo tracklet file formats)."""
import math
import warnings

import numpy as np
import torch
import torch.nn.functional as F

from .registry import PIPELINES


def _points_list(results):
    p = results['points']
    return p if isinstance(p, (list, tuple)) else [p]


def _apply_points(results, fn):
    p = results['points']
    if isinstance(p, (list, tuple)):
        for t in p:
            fn(t)
    else:
        fn(p)


def _candidates(results):
    return results.get('gt_tracklet_candidates', [])


@PIPELINES.register_module()
class LoadTrackletPoints(object):
    """The points of one tracklet: a .npy holding an object array of per-frame [n_i, load_dim] float32 arrays
    (tracklet_pipelines.py:26-91); keeps the first use_dim columns, caps every frame at max_points (random subset)."""

    def __init__(self, load_dim=5, use_dim=5, coord_type='LIDAR', max_points=-1, debug=False):
        self.load_dim, self.use_dim, self.coord_type, self.max_points, self.debug = load_dim, use_dim, coord_type, max_points, debug

    def __call__(self, results):
        trk = results['tracklet']
        if self.debug:
            pts = [np.random.rand(100, 6).astype(np.float32) * 2 for _ in range(len(trk))]
            for i, p in enumerate(pts):
                p[:, :3] += trk.boxes[i:i + 1, :3].float().cpu().numpy()
        else:
            pts = np.load(results['pts_filename'], allow_pickle=True)
        if results.get('point_cloud_interval', None) is not None:
            beg, end = results['point_cloud_interval']
            pts = pts[beg:end]
        assert len(pts) == len(trk)
        assert self.load_dim == pts[0].shape[1]
        pts = [torch.from_numpy(np.ascontiguousarray(p[:, :self.use_dim])) for p in pts]
        frames = [torch.ones(len(p), dtype=torch.int) * i for i, p in enumerate(pts)]
        if self.max_points > 0:
            pts, frames = self.points_downsample(pts, frames)
        results['points'], results['pts_frame_inds'] = pts, frames
        return results

    def points_downsample(self, pts, frames):
        out_p, out_f = [], []
        for p, f in zip(pts, frames):
            if len(p) > self.max_points:
                idx = torch.randperm(len(p))[:self.max_points]
                p, f = p[idx], f[idx]
            out_p.append(p)
            out_f.append(f)
        return out_p, out_f


@PIPELINES.register_module()
class LoadTrackletAnnotations(object):
    """tracklet_pipelines.py:94-101."""

    def __call__(self, results):
        results['gt_tracklet_candidates'] = results['ann_info']
        return results


@PIPELINES.register_module()
class LoadAnnotationsOcc(object):
    """The occupancy label grid of every GT candidate: <occ_anno_root>/<segment>/<track id>.npz, key 'occ', an
    X x Y x Z integer grid in {0 unknown, 1 occupied, 2 free} (occ_pinelines.py:33-80); a missing file or name gives
    a 1x1x1 unknown grid with score 0."""

    def __init__(self, compute_score=False):
        self.compute_score = compute_score

    def __call__(self, results):
        grids, scores, lengths = [], [], []
        for info in results['occ_infos']:
            score, length = info['label_iou'], info['label_trk_length']
            if info['occ_label_name'] is None:
                grids.append(torch.zeros(1, 1, 1, dtype=torch.int))
                score = 0.0
            else:
                try:
                    occ = torch.from_numpy(np.load(info['occ_label_name'])['occ'])
                    grids.append(occ)
                    if self.compute_score:
                        score = (occ.numel() - int((occ == 0).sum())) / occ.numel()
                except FileNotFoundError:
                    grids.append(torch.zeros(1, 1, 1, dtype=torch.int))
                    score = 0.0
            scores.append(score)
            lengths.append(length)
        if 'gt_bboxes_3d' in results and len(results['gt_bboxes_3d']) > len(grids):
            for _ in range(len(results['gt_bboxes_3d']) - len(grids)):
                grids.append(torch.zeros(1, 1, 1, dtype=torch.int))
                scores.append(0.0)
                lengths.append(0)
        results['occ_label_list'] = grids
        results['occ_scores'] = torch.tensor(scores)
        results['occ_lengths'] = torch.tensor(lengths, dtype=torch.int)
        return results


@PIPELINES.register_module()
class TrackletRegularization(object):
    """Pad (repeat the last frame) or cut (random head / tail) the tracklet to reg_len frames."""

    def __init__(self, reg_len=150):
        self.reg_len = reg_len

    def __call__(self, results):
        trk = results['tracklet']
        if len(trk) == self.reg_len:
            return results
        points, frames = list(results['points']), list(results['pts_frame_inds'])
        if len(trk) < self.reg_len:
            warnings.warn(f'tracklet length {len(trk)} < {self.reg_len}')
            pad = self.reg_len - len(trk)
            trk.boxes = torch.cat([trk.boxes, trk.boxes[-1:].expand(pad, -1)], 0)
            trk.scores = torch.cat([trk.scores, trk.scores[-1:].expand(pad)], 0)
            trk.ts_list = trk.ts_list + [trk.ts_list[-1]] * pad
            if getattr(trk, 'pose_list', None) is not None:
                trk.pose_list = list(trk.pose_list) + [trk.pose_list[-1]] * pad
            points = points + [points[-1]] * pad
            frames = frames + [frames[-1]] * pad
        else:
            cut = len(trk) - self.reg_len
            head = np.random.randint(0, cut)
            tail = cut - head
            points = points[head:-tail]
            frames = [torch.ones(len(p), dtype=torch.int) * i for i, p in enumerate(points)]
            trk.select(range(head, len(trk) - tail))
        trk.ts2index = {ts: i for i, ts in enumerate(trk.ts_list)}
        results['points'], results['pts_frame_inds'] = points, frames
        assert len(points) == len(trk)
        return results


@PIPELINES.register_module()
class FrameDropout(object):
    def __init__(self, drop_ratio=0.1):
        self.drop_ratio = drop_ratio

    def __call__(self, results):
        trk = results['tracklet']
        n = len(trk)
        # LiDARTracklet.random_frame_drop (lidar_tracklet.py:120-128): np.random.choice WITH replacement over the
        # timestamps, so up to drop_num distinct frames go
        num_drop = int(n * self.drop_ratio)
        if n - num_drop <= 0:
            keep = list(range(n))
        else:
            drop = set(np.random.choice(trk.ts_list, num_drop).tolist())
            keep = [i for i, ts in enumerate(trk.ts_list) if ts not in drop]
        trk.select(keep)
        results['points'] = [results['points'][i] for i in keep]
        results['pts_frame_inds'] = [results['pts_frame_inds'][i] for i in keep]
        return results


@PIPELINES.register_module()
class TrackletNoise(object):
    """Uniform noise on centres (additive), sizes (multiplicative, 1 +- max) and yaw, per frame or one draw for
    the whole tracklet ('consistent'); torch.rand draws in the reference's order and shapes."""

    def __init__(self, center_noise_cfg=None, size_noise_cfg=None, yaw_noise_cfg=None):
        self.c_cfg, self.s_cfg, self.y_cfg = center_noise_cfg, size_noise_cfg, yaw_noise_cfg

    @staticmethod
    def _rand(shape, like):
        return torch.rand(shape, dtype=like.dtype, device=like.device)

    def __call__(self, results):
        trk = results['tracklet']
        b = trk.boxes
        if len(trk) == 0:
            return results
        if self.c_cfg is not None:
            mx = b.new_tensor(self.c_cfg['max_noise'])
            assert mx.numel() == 3
            noise = (self._rand(3, b) - 0.5) * 2 * mx if self.c_cfg['consistent'] else \
                (self._rand((len(trk), 3), b) - 0.5) * 2 * mx[None, :]
            b[:, :3] += noise
        if self.s_cfg is not None:
            mx = b.new_tensor(self.s_cfg['max_noise'])
            assert mx.numel() == 3 and bool((mx < 0.5).all())
            noise = 1 + (self._rand(3, b) - 0.5) * 2 * mx if self.s_cfg['consistent'] else \
                1 + (self._rand((len(trk), 3), b) - 0.5) * 2 * mx[None, :]
            b[:, 3:6] *= noise
        if self.y_cfg is not None:
            mx = self.y_cfg['max_noise']
            noise = (self._rand(1, b) - 0.5) * 2 * mx if self.y_cfg['consistent'] else \
                (self._rand(len(trk), b) - 0.5) * 2 * mx
            b[:, 6] += noise
        return results


@PIPELINES.register_module()
class PointDecoration(object):
    """Append per-frame box attributes to every point of the frame: yaw / 3.1415, size / 10, score,
    (xyz - box centre) / 5, tracklet length / 100 -- the 'pts_feats' columns the RoI head reads."""

    def __init__(self, properties, concat=True):
        self.properties, self.concat = properties, concat

    def __call__(self, results):
        trk = results['tracklet']
        points = list(results['points'])
        assert len(points) == len(trk), f'{len(points)}, {len(trk)}'
        for pro in self.properties:
            points = getattr(self, pro)(points, trk)
        if self.concat:
            results['points'] = torch.cat(points, 0)
            results['pts_frame_inds'] = torch.cat(list(results['pts_frame_inds']))
        else:
            results['points'] = points
        return results

    @staticmethod
    def yaw(points, trk):
        return [F.pad(p, (0, 1), 'constant', float(trk.boxes[i, 6]) / 3.1415) for i, p in enumerate(points)]

    @staticmethod
    def size(points, trk):
        return [torch.cat([p, (trk.boxes[i:i + 1, 3:6] / 10).to(p).expand(len(p), -1)], 1) for i, p in enumerate(points)]

    @staticmethod
    def score(points, trk):
        return [F.pad(p, (0, 1), 'constant', float(trk.scores[i])) for i, p in enumerate(points)]

    @staticmethod
    def center_offset(points, trk):
        return [torch.cat([p, (p[:, :3] - trk.boxes[i:i + 1, :3].to(p)) / 5], 1) for i, p in enumerate(points)]

    @staticmethod
    def length(points, trk):
        return [F.pad(p, (0, 1), 'constant', len(trk) / 100) for p in points]


@PIPELINES.register_module()
class TrackletRandomFlip(object):
    """Flip points, the tracklet and the GT candidates along the BEV axes; records pcd_horizontal_flip /
    pcd_vertical_flip (what TrackletRoIHeadOCC.inverse_aug reads at test time)."""

    def __init__(self, flip_ratio_bev_horizontal=0.0, flip_ratio_bev_vertical=0.0, **kwargs):
        for r in (flip_ratio_bev_horizontal, flip_ratio_bev_vertical):
            assert r is None or (isinstance(r, (int, float)) and 0 <= r <= 1)
        self.flip_ratio_bev_horizontal, self.flip_ratio_bev_vertical = flip_ratio_bev_horizontal, flip_ratio_bev_vertical

    @staticmethod
    def _flip(results, direction):
        col = 1 if direction == 'horizontal' else 0

        def f(p):
            p[:, col] = -p[:, col]
        _apply_points(results, f)
        results['tracklet'].flip(direction)
        for t in _candidates(results):
            t.flip(direction)

    def __call__(self, results):
        if 'pcd_horizontal_flip' not in results:
            results['pcd_horizontal_flip'] = bool(np.random.rand() < self.flip_ratio_bev_horizontal)
        if 'pcd_vertical_flip' not in results:
            results['pcd_vertical_flip'] = bool(np.random.rand() < self.flip_ratio_bev_vertical)
        if results['pcd_horizontal_flip']:
            self._flip(results, 'horizontal')
        if results['pcd_vertical_flip']:
            self._flip(results, 'vertical')
        return results


@PIPELINES.register_module()
class TrackletPoseTransform(object):
    """Bring every frame's points and boxes from that frame's ego pose into the ego frame of the tracklet's middle
    frame (tracklet_pipelines.py:228-303).  tracklet.pose_list holds the per-frame ego -> world 4x4 poses."""

    def __init__(self, concat=True, centering=False):
        self.concat, self.centering = concat, centering

    @staticmethod
    def points_frame_transform(src_points, src_pose, tgt_pose, tgt_pose_inv=None):
        h = F.pad(src_points, (0, 1), 'constant', 1)
        world2tgt = torch.inverse(tgt_pose) if tgt_pose_inv is None else tgt_pose_inv
        mm = world2tgt @ src_pose
        return (h @ mm.T.to(h.dtype))[:, :3]

    def __call__(self, results):
        points, trk = list(results['points']), results['tracklet']
        poses = trk.pose_list
        assert getattr(trk, 'shared_pose', None) is None
        assert len(points) == len(trk) == len(poses)
        center_pose = poses[len(poses) // 2]
        trk.frame_transform(center_pose)
        for t in _candidates(results):
            t.frame_transform(center_pose)
        inv = torch.linalg.inv(center_pose)
        points = [torch.cat([self.points_frame_transform(p[:, :3], pose, None, inv), p[:, 3:]], 1)
                  for pose, p in zip(poses, points)]
        if self.centering:  # translation only: the middle frame's box centre becomes the origin
            translation = -1 * trk.boxes[len(trk) // 2:len(trk) // 2 + 1, :3].clone()
            for p in points:
                p[:, :3] += translation.to(p)
            trk.translate(translation)
            for t in _candidates(results):
                t.translate(translation)
            trk.translation_factor = translation.cpu().numpy()
        results['shared_pose'] = center_pose
        if self.concat:
            results['points'] = torch.cat(points, 0)
            results['pts_frame_inds'] = torch.cat(list(results['pts_frame_inds']))
        else:
            results['points'] = points
        return results


@PIPELINES.register_module()
class TrackletGlobalRotScaleTrans(object):
    """Random rotation about z, isotropic scaling and Gaussian translation of points, tracklet and GT candidates;
    records pcd_rot_angle / pcd_scale_factor / pcd_trans (and tracklet.rot_angle)."""

    def __init__(self, rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05),
                 translation_std=(0, 0, 0), shift_height=False):
        if isinstance(rot_range, (int, float)):
            rot_range = [-rot_range, rot_range]
        self.rot_range = rot_range
        assert isinstance(scale_ratio_range, (list, tuple))
        self.scale_ratio_range = scale_ratio_range
        if not isinstance(translation_std, (list, tuple, np.ndarray)):
            translation_std = [translation_std] * 3
        assert all(s >= 0 for s in translation_std)
        self.translation_std = translation_std
        self.shift_height = shift_height

    def __call__(self, results):
        trk, cands = results['tracklet'], _candidates(results)
        if 'pcd_rot_angle' not in results:
            results['pcd_rot_angle'] = np.random.uniform(self.rot_range[0], self.rot_range[1])
        angle = results['pcd_rot_angle']
        trk.rotate(angle)
        trk.rot_angle = angle
        for t in cands:
            t.rotate(angle)

        def rot(p):  # BasePoints.rotate(-angle): xyz @ [[c,-s,0],[s,c,0],[0,0,1]]^T with c, s of -angle
            a = p.new_tensor(-angle)
            s, c = torch.sin(a), torch.cos(a)
            m = p.new_tensor([[c, -s, 0.], [s, c, 0.], [0., 0., 1.]]).T
            p[:, :3] = p[:, :3] @ m
        _apply_points(results, rot)

        if 'pcd_scale_factor' not in results:
            results['pcd_scale_factor'] = np.random.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1])
        scale = results['pcd_scale_factor']

        def sc(p):
            p[:, :3] *= scale
        _apply_points(results, sc)
        trk.scale(scale)
        for t in cands:
            t.scale(scale)

        trans = np.random.normal(scale=np.array(self.translation_std, dtype=np.float32), size=3).T
        results['pcd_trans'] = trans

        def tr(p):
            p[:, :3] += p.new_tensor(trans)
        _apply_points(results, tr)
        trk.translate(trans)
        for t in cands:
            t.translate(trans)
        return results


@PIPELINES.register_module()
class PointsRangeFilter(object):
    """Keep the points inside point_cloud_range (strict inequalities, as BasePoints.in_range_3d)."""

    def __init__(self, point_cloud_range):
        self.pcd_range = np.array(point_cloud_range, dtype=np.float32)

    def __call__(self, results):
        r = self.pcd_range

        def mask(p):
            return ((p[:, 0] > r[0]) & (p[:, 1] > r[1]) & (p[:, 2] > r[2]) &
                    (p[:, 0] < r[3]) & (p[:, 1] < r[4]) & (p[:, 2] < r[5]))
        p = results['points']
        if isinstance(p, (list, tuple)):
            ms = [mask(t) for t in p]
            results['points'] = [t[m] for t, m in zip(p, ms)]
            results['pts_frame_inds'] = [f[m] for f, m in zip(results['pts_frame_inds'], ms)]
        else:
            m = mask(p)
            results['points'] = p[m]
            if 'pts_frame_inds' in results:
                results['pts_frame_inds'] = results['pts_frame_inds'][m]
        return results


@PIPELINES.register_module()
class PointShuffle(object):
    def __call__(self, results):
        p = results['points']
        assert not isinstance(p, (list, tuple)), 'shuffle after the concatenating step'
        idx = torch.randperm(p.shape[0], device=p.device)
        results['points'] = p[idx]
        if 'pts_frame_inds' in results:
            results['pts_frame_inds'] = results['pts_frame_inds'][idx]
        return results


class Compose(object):
    """mmdet Compose: a list of transform configs (dicts with 'type') or callables."""

    def __init__(self, transforms):
        self.transforms = [PIPELINES.build(t) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, results):
        for t in self.transforms:
            results = t(results)
            if results is None:
                return None
        return results


# ---- occupancy-label transforms (mmdet3d/datasets/pipelines/occ_pinelines.py) ----
def _grid_coors(shape):
    xs, ys, zs = shape
    gx, gy, gz = torch.meshgrid(torch.arange(xs, dtype=torch.long), torch.arange(ys, dtype=torch.long),
                                torch.arange(zs, dtype=torch.long), indexing='ij')
    return gx, gy, gz


def _mirror_fill(occ_grid):
    """Unknown cells (0) take the label of the cell mirrored along x (occ_pinelines.py:96-121 / 209-224)."""
    xs = occ_grid.shape[0]
    flat = occ_grid.clone().view(-1)
    unknown = flat == 0
    gx, gy, gz = _grid_coors(occ_grid.shape)
    mid = xs // 2
    mx = ((gx + 0.5 - mid) * -1.0 + mid).long()
    mc = torch.stack([mx, gy, gz], -1).view(-1, 3)[unknown]
    flat[unknown] = occ_grid[mc[:, 0], mc[:, 1], mc[:, 2]]
    return flat.view(occ_grid.shape)


@PIPELINES.register_module()
class MirrorOccLabel(object):
    """occ_pinelines.py:83-127."""

    def __call__(self, results):
        if 'occ_label_list' in results:
            results['occ_label_list'] = [_mirror_fill(g) for g in results['occ_label_list']]
        return results


@PIPELINES.register_module()
class RandomSampleOccPoints(object):
    """Sample the K occupancy query points of every annotated object grid (occ_pinelines.py:130-359): labels
    {0 unknown, 1 occupied, 2 free} on an X x Y x Z grid -> sample_occs [N,K], sample_occ_centers [N,K,3]
    (cell centres in the box frame, origin at the grid's centre), occ_sizes [N,3].  Same branches and the same
    torch.multinomial / topk call order as the reference, so the same seed gives the same samples."""

    def __init__(self, num_sample_points=1024, pos_sample_weight=0.5, voxel_size=0.2, use_unknown=False,
                 use_potential=False, mirror_x=False, balance_sample=False, weighted_sample=True):
        self.num_sample_points, self.pos_sample_weight, self.voxel_size = num_sample_points, pos_sample_weight, voxel_size
        self.use_unknown, self.use_potential = use_unknown, use_potential
        if use_potential:
            self.potential = {}
        self.mirror_x, self.balance_sample, self.weighted_sample = mirror_x, balance_sample, weighted_sample

    def __call__(self, results):
        if 'occ_label_list' not in results:
            return results
        infos, grids, scores = results['occ_infos'], results['occ_label_list'], results['occ_scores']
        k_fixed = 0 if self.num_sample_points == -1 else self.num_sample_points
        if len(grids) == 0:
            results['sample_occs'] = torch.zeros((0, k_fixed))
            results['sample_occ_centers'] = torch.zeros((0, k_fixed, 3))
            results['occ_sizes'] = torch.zeros((0, 3))
            return results
        out_occ, out_ctr, out_size = [], [], []
        K = self.num_sample_points
        for i, (grid, score, info) in enumerate(zip(grids, scores, infos)):
            if not bool((grid > 0).any()):  # nothing annotated: an empty sample of the right shape
                assert score == 0, 'occ_score should be 0 if no occ grid is annotated'
                ctr, occ = torch.zeros(k_fixed, 3), torch.zeros(k_fixed)
                w = l = h = 0.0
            else:
                xs, ys, zs = grid.shape
                flat = grid.view(-1)
                gx, gy, gz = _grid_coors(grid.shape)
                if self.mirror_x:
                    filled = _mirror_fill(grid).view(-1)
                    flat[flat == 0] = filled[flat == 0]     # (the reference writes through the view into the grid)
                coors = torch.stack([gx, gy, gz], -1).view(-1, 3)
                if not self.use_unknown:
                    valid_coors, valid = coors[flat > 0], flat[flat > 0]
                else:
                    valid_coors, valid = coors, flat.clone()
                w, l, h = float(xs) * self.voxel_size, float(ys) * self.voxel_size, float(zs) * self.voxel_size
                min_bound = torch.tensor([-w / 2, -l / 2, -h / 2], dtype=torch.float32)
                centers = valid_coors.to(torch.float) * self.voxel_size + min_bound + self.voxel_size / 2
                if K == -1:
                    idx = torch.arange(len(centers))
                elif self.balance_sample:
                    num_pos = int(K * self.pos_sample_weight)
                    num_neg = K - num_pos
                    ids = torch.arange(len(valid))
                    pos, neg = ids[valid == 1], ids[valid != 1]
                    if len(pos) == 0 or len(neg) == 0:
                        idx = torch.multinomial(torch.ones_like(valid, dtype=torch.float), K, replacement=len(valid) < K)
                        scores[i] = 0.0  # do not use this sample
                    else:
                        pc = torch.multinomial(torch.ones_like(pos, dtype=torch.float), num_pos, replacement=len(pos) < num_pos)
                        nc = torch.multinomial(torch.ones_like(neg, dtype=torch.float), num_neg, replacement=len(neg) < num_neg)
                        idx = torch.cat([pos[pc], neg[nc]], 0)
                elif self.use_potential:
                    pot = self.potential.get(info['occ_label_name'], torch.ones_like(valid, dtype=torch.float))
                    if len(valid) < K:
                        idx = torch.multinomial(1 / pot, K, replacement=True)
                    else:
                        _, idx = torch.topk(pot, K, dim=0, largest=False)
                    pot[idx] += 1
                    self.potential[info['occ_label_name']] = pot
                elif self.weighted_sample:
                    wts = torch.ones_like(valid, dtype=torch.float) * (1 - self.pos_sample_weight)
                    wts[valid == 1] = self.pos_sample_weight
                    if float(wts.sum()) <= 0:  # (the reference lands here through an exception)
                        wts = torch.ones_like(valid, dtype=torch.float)
                    idx = torch.multinomial(wts, K, replacement=len(valid) < K)
                else:
                    idx = torch.multinomial(torch.ones_like(valid, dtype=torch.float), K, replacement=len(valid) < K)
                ctr, occ = centers[idx], valid[idx]
            out_occ.append(occ)
            out_ctr.append(ctr)
            out_size.append(torch.tensor([w, l, h], dtype=torch.float32))
        if K != -1:
            results['sample_occs'] = torch.stack(out_occ, 0)
            results['sample_occ_centers'] = torch.stack(out_ctr, 0)
        else:
            results['sample_occs'], results['sample_occ_centers'] = out_occ, out_ctr
        results['occ_sizes'] = torch.stack(out_size, 0)
        return results


@PIPELINES.register_module()
class JitterOccCenter(object):
    """Move every sampled centre uniformly inside its cell (occ_pinelines.py:362-377)."""

    def __init__(self, voxel_size=0.2):
        self.voxel_size = voxel_size

    def __call__(self, results):
        c = results['sample_occ_centers']
        results['sample_occ_centers'] = c + (torch.rand_like(c) * self.voxel_size - self.voxel_size / 2)
        return results


@PIPELINES.register_module()
class TrackletOccFormatBundle(object):
    """occ_labels [N,K,4] = sampled centres + label, occ_labels_scores (formating.py:337-355; the DataContainer
    wrappers of mmcv are not needed without its collate)."""

    def __init__(self, class_names=None, **kwargs):
        self.class_names = class_names

    def __call__(self, results):
        if 'sample_occ_centers' in results and 'sample_occs' in results:
            c, o = results['sample_occ_centers'], results['sample_occs']
            if isinstance(c, list):
                results['occ_labels'] = [torch.cat([ci, oi.unsqueeze(-1).to(ci)], -1) for ci, oi in zip(c, o)]
            else:
                results['occ_labels'] = torch.cat([c, o.unsqueeze(-1).to(c)], -1)
        if 'occ_scores' in results:
            results['occ_labels_scores'] = results['occ_scores']
        return results


@PIPELINES.register_module()
class Collect3D(object):
    def __init__(self, keys, meta_keys=('pcd_horizontal_flip', 'pcd_vertical_flip', 'pcd_rot_angle', 'pcd_scale_factor',
                                         'pcd_trans', 'sample_idx', 'pts_filename')):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        out = {k: results[k] for k in self.keys}
        out['img_metas'] = {k: results[k] for k in self.meta_keys if k in results}
        return out


def collate_tracklets(samples, device):
    """A batch of pipeline outputs -> the keyword arguments of TrackletDetectorOCC.forward (what mmcv's collate +
    scatter hand the reference model, tracklet_detector_occ.py:96-150): per-sample lists, tensors on ``device``."""
    def trk_to(t):
        t = t.clone()
        t.boxes, t.scores = t.boxes.to(device), t.scores.to(device)
        return t
    batch = dict(points=[s['points'].to(device) for s in samples],
                 pts_frame_inds=[s['pts_frame_inds'].to(device).long() for s in samples],
                 img_metas=[s.get('img_metas', {}) for s in samples],
                 tracklet=[trk_to(s['tracklet']) for s in samples])
    if 'gt_tracklet_candidates' in samples[0]:
        batch['gt_tracklet_candidates'] = [[trk_to(c) for c in s['gt_tracklet_candidates']] for s in samples]
    if 'occ_labels' in samples[0]:
        batch['occ_labels'] = [[o.to(device).float() for o in s['occ_labels']] for s in samples]
        batch['occ_labels_scores'] = [[sc.reshape(1).to(device).float() for sc in s['occ_labels_scores']] for s in samples]
    return batch
