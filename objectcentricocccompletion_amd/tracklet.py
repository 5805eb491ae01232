"""Tracklet container and assignment -- the parts of LiDARTracklet
(mmdet3d/core/bbox/structures/lidar_tracklet.py:253-339, 552-607) and TrackletAssigner
(mmdet3d/core/bbox/assigners/tracklet_assigner.py:14-57) the RoI head and the test-time
augmentation touch, on plain tensors: boxes [L,7] (x,y,z_bottom,w,l,h,yaw), timestamps,
per-box scores, a class id."""
import math

import torch

from . import _lib as L
from .registry import BBOX_ASSIGNERS


def aligned_iou_3d(boxes1, boxes2):
    """i-th box of boxes1 with i-th box of boxes2 (lidar_box3d.py:404-448), HIP kernel."""
    L.require_device(boxes1, boxes2)
    b1, b2 = boxes1[:, :7].contiguous().float(), boxes2[:, :7].contiguous().float()
    assert b1.shape == b2.shape
    out = torch.empty((b1.size(0),), dtype=torch.float32, device=b1.device)
    L.check(L.lib.ococc_aligned_iou3d_f32(L.ptr(b1), L.ptr(b2), b1.size(0), L.ptr(out), L.stream()),
            'aligned_iou3d')
    return out


def host_index(values, device, dtype=torch.long):
    """A small index list built on the host -> device tensor WITHOUT a device synchronisation (torch.tensor(list,
    device=...) is a pageable copy and waits for the stream): staged through pinned memory, copied asynchronously."""
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type == 'cpu' or t.numel() == 0:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def host_index_many(lists, device, dtype=torch.long):
    """Several index lists in ONE pinned staging buffer and ONE asynchronous copy: views of the device buffer, in order."""
    sizes = [len(v) for v in lists]
    flat = host_index([x for v in lists for x in v], device, dtype)
    return flat.split(sizes) if sizes else ()


def _median_lower_upper_mean(x):
    """numpy.median along dim 0 (for an even count: the mean of the two middle values; torch.median
    would return the lower one)."""
    v = torch.sort(x, 0).values
    n = v.size(0)
    return v[n // 2] if n % 2 == 1 else 0.5 * (v[n // 2 - 1] + v[n // 2])


class Tracklet(object):
    """Boxes of one object over time.  ``type`` is the class id (0 = vehicle in the ococcnet
    config)."""

    def __init__(self, boxes, ts_list, scores=None, type=0, segment_name=None, id=None):
        self.boxes = boxes
        self.ts_list = list(ts_list)
        assert len(self.ts_list) == boxes.size(0)
        self.scores = scores if scores is not None else boxes.new_ones((boxes.size(0),))
        self.type = type
        self.segment_name, self.id = segment_name, id
        self.ts2index = {ts: i for i, ts in enumerate(self.ts_list)}

    def __len__(self):
        return self.boxes.size(0)

    @property
    def device(self):
        return self.boxes.device

    def new_empty(self):
        return Tracklet(self.boxes.new_zeros((0, self.boxes.size(1))), [], self.scores.new_zeros((0,)), self.type)

    def concated_boxes(self):
        return self.boxes

    def concated_scores(self):
        return self.scores

    def concated_labels(self):
        return torch.full((len(self),), self.type, device=self.device, dtype=torch.long)

    def get_index_from_ts(self, ts):
        return self.ts2index.get(ts, -1)

    def ts_intersection(self, trk):
        other = set(trk.ts_list)
        return [ts for ts in self.ts_list if ts in other]

    def common_frames(self, trk):
        """Host-side matching of timestamps: (own frame indices, the other tracklet's frame indices)."""
        inter = self.ts_intersection(trk)
        return [self.ts2index[t] for t in inter], [trk.ts2index[t] for t in inter]

    def intersection_ious(self, trk):
        """IoU of the boxes the two tracklets have at common timestamps (lidar_tracklet.py:290-299)."""
        i1, i2 = self.common_frames(trk)
        if len(i1) == 0:
            return self.boxes.new_zeros(0)
        if i1 == i2 and len(i1) == len(self) == len(trk):   # same frames in the same order: no gather at all
            return aligned_iou_3d(self.boxes, trk.boxes)
        return aligned_iou_3d(self.boxes[host_index(i1, self.device)], trk.boxes[host_index(i2, self.device)])

    def self_ious(self, trk):
        """Per own box: IoU with the other tracklet's box of the same timestamp, 0 if none (:278-288).  The RoI head
        computes these for a whole batch in one launch and leaves them in ``_self_iou_cache`` (roi_head.py)."""
        hit = getattr(self, '_self_iou_cache', None)
        if hit is not None:
            cand, boxes, over = hit[:3]
            # the very candidate object, and neither side transformed since (the in-place transforms bump _version)
            if cand is trk and boxes is self.boxes and hit[3:] == (self.boxes._version, trk.boxes._version):
                return over
            self._self_iou_cache = None
        out = self.boxes.new_zeros(len(self))
        i1, _ = self.common_frames(trk)
        if len(i1) == 0:
            return out
        if len(i1) == len(self):
            return self.intersection_ious(trk) if i1 == list(range(len(self))) else out.index_copy(0, host_index(i1, self.device), self.intersection_ious(trk))
        out[host_index(i1, self.device)] = self.intersection_ious(trk)
        return out

    # ---- in-place geometric transforms: LiDARTracklet.flip / translate / scale / rotate
    # (lidar_tracklet.py:253-276) applying LiDARInstance3DBoxes.flip / rotate (lidar_box3d.py:143-216)
    # and BaseInstance3DBoxes.translate / scale (base_box3d.py:156-231) to every box ----
    def flip(self, direction):
        assert direction in ('horizontal', 'vertical')
        if direction == 'horizontal':  # y -> -y, yaw -> pi - yaw
            self.boxes[:, 1] = -self.boxes[:, 1]
            self.boxes[:, 6] = -self.boxes[:, 6] + math.pi
        else:                          # x -> -x, yaw -> -yaw
            self.boxes[:, 0] = -self.boxes[:, 0]
            self.boxes[:, 6] = -self.boxes[:, 6]

    def translate(self, trans):
        self.boxes[:, :3] += torch.as_tensor(trans, dtype=self.boxes.dtype, device=self.device).view(-1)[:3]

    def scale(self, scale):
        self.boxes[:, :6] *= scale

    def rotate(self, angle):
        a = torch.as_tensor(angle, dtype=self.boxes.dtype, device=self.device)
        s, c = torch.sin(a), torch.cos(a)
        rot_t = self.boxes.new_tensor([[c, -s, 0.], [s, c, 0.], [0., 0., 1.]])
        self.boxes[:, :3] = self.boxes[:, :3] @ rot_t
        self.boxes[:, 6] += a

    # ---- on-disk form (LiDARTracklet.to_dump_format / from_dump_format, lidar_tracklet.py:130-161): the tuple
    # (segment_name, id, type, in_world, [box [1,7] ndarray per frame], ts_list, score_list, num_pts_in_boxes)
    # that the *_training.pkl proposal files and *_gt_candidates.pkl annotation files hold ----
    type_mapping = {1: 'Car', 2: 'Pedestrian', 4: 'Cyclist'}  # Waymo label ids

    @classmethod
    def from_dump_format(cls, item, device='cpu'):
        seg, id_, type_, in_world, boxes, ts_list, scores, num_pts = item
        import numpy as np
        b = torch.from_numpy(np.concatenate([np.asarray(x, dtype=np.float32).reshape(1, -1) for x in boxes], 0)) \
            if len(boxes) else torch.zeros((0, 7))
        assert list(ts_list) == sorted(ts_list) and len(set(ts_list)) == len(ts_list)
        t = cls(b.to(device), list(ts_list), torch.as_tensor(list(scores), dtype=torch.float32, device=device), type_, seg, id_)
        t.in_world, t.num_pts_in_boxes, t.type_format = in_world, num_pts, 'waymo'
        return t

    def to_dump_format(self):
        b = self.boxes.detach().cpu().numpy()
        return (self.segment_name, self.id, self.type, getattr(self, 'in_world', False), [b[i:i + 1] for i in range(len(b))],
                list(self.ts_list), [float(s) for s in self.scores], getattr(self, 'num_pts_in_boxes', None))

    def set_poses(self, ts2poses):
        self.pose_list = [ts2poses[ts] for ts in self.ts_list]

    def set_type_name(self):
        assert getattr(self, 'type_format', 'waymo') == 'waymo'
        self.type_name = self.type_mapping[self.type]

    def set_type(self, type, format):
        self.type, self.type_format = type, format

    def frame_transform(self, pose):
        """Every frame's box from that frame's ego pose (self.pose_list[i], ego -> world 4x4) into the frame of
        ``pose`` (LiDARTracklet.frame_transform, lidar_tracklet.py:348-387): centres through
        inv(pose) @ pose_i, the yaw from the transformed heading vector (sin, cos, 0)."""
        assert getattr(self, 'shared_pose', None) is None and len(self.pose_list) == len(self)
        world2tgt = torch.linalg.inv(pose)
        out = self.boxes.clone()
        for i in range(len(self)):
            mm = (world2tgt @ self.pose_list[i]).to(self.boxes.dtype)
            c = torch.cat([self.boxes[i, :3], self.boxes.new_ones(1)])
            out[i, :3] = (mm @ c)[:3]
            yaw = self.boxes[i, 6]
            hv = torch.stack([torch.sin(yaw), torch.cos(yaw), yaw.new_zeros(()), yaw.new_ones(())])
            mm = mm.clone()
            mm[:3, 3] = 0
            t = mm @ hv
            out[i, 6] = torch.atan2(t[0], t[1])
        self.boxes = out
        self.shared_pose = pose

    def shared2ego(self, boxes=None):
        """Boxes expressed in the tracklet's shared frame back into each frame's own ego frame
        (LiDARTracklet.shared2ego, lidar_tracklet.py:449-492): centres through inv(pose_i) @ shared_pose, the yaw
        from the transformed heading vector (sin, cos, 0).  The 4x4 inverses are taken on the host (f32 poses, a
        handful per tracklet), as the reference does."""
        src = self.boxes if boxes is None else boxes
        tgt_pose = torch.stack([p.to(self.device) for p in self.pose_list], 0)
        mm = torch.linalg.inv(tgt_pose.cpu()).to(self.device) @ self.shared_pose.to(self.device)
        ones = src.new_ones((src.size(0), 1))
        centre = torch.einsum('nij,nj->ni', mm.to(src.dtype), torch.cat([src[:, :3], ones], 1))[:, :3]
        rot = mm.clone().to(src.dtype)
        rot[:, :3, 3] = 0
        hv = torch.stack([torch.sin(src[:, 6]), torch.cos(src[:, 6]), torch.zeros_like(src[:, 6])], 1)
        hv = torch.einsum('nij,nj->ni', rot, torch.cat([hv, ones], 1))[:, :3]
        return torch.cat([centre, src[:, 3:6], torch.atan2(hv[:, 0], hv[:, 1])[:, None]], 1)

    def update_from_prediction(self, boxes, scores, labels, valid_mask, to_ego=True):
        """Take over refined boxes / scores (LiDARTracklet.update_from_prediction, lidar_tracklet.py:403-447): frames
        whose RoI was empty (valid_mask False) keep their own box and score; with to_ego both are expressed in the
        per-frame ego frames, as the evaluation wants; an augmentation translation recorded on the tracklet is undone
        first.  A tracklet without poses is already in one frame and is left there."""
        assert len(boxes) == len(scores) == len(labels) == len(valid_mask) == len(self)
        assert bool((labels == labels[0]).all())
        self.type = int(labels[0])
        boxes = boxes[:, :7].clone()
        if getattr(self, 'translation_factor', None) is not None:
            back = -torch.as_tensor(self.translation_factor, dtype=boxes.dtype, device=self.device).view(-1)[:3]
            boxes[:, :3] += back
            self.translate(back)
        posed = getattr(self, 'pose_list', None) is not None and getattr(self, 'shared_pose', None) is not None
        new = self.shared2ego(boxes) if (to_ego and posed) else boxes
        old = self.shared2ego() if posed else self.boxes[:, :7]
        keep = valid_mask.to(self.device).bool()
        self.boxes = torch.where(keep[:, None], new.to(self.boxes.dtype), old)
        self.scores = torch.where(keep, scores.to(self.scores.dtype), self.scores)
        self.pose_list = None

    def select(self, keep):
        """Keep the frames with the given positions (LiDARTracklet.remove over its list fields, lidar_tracklet.py:106-118)."""
        keep = list(keep)
        idx = torch.as_tensor(keep, dtype=torch.long, device=self.device)
        self.boxes, self.scores = self.boxes[idx], self.scores[idx]
        self.ts_list = [self.ts_list[i] for i in keep]
        if getattr(self, 'pose_list', None) is not None:
            self.pose_list = [self.pose_list[i] for i in keep]
        self.ts2index = {ts: i for i, ts in enumerate(self.ts_list)}

    def clone(self):
        t = Tracklet(self.boxes.clone(), list(self.ts_list), self.scores.clone(), self.type, self.segment_name, self.id)
        for k in ('rot_angle', 'pose_list', 'shared_pose', 'in_world', 'num_pts_in_boxes', 'type_format', 'type_name'):
            if hasattr(self, k):
                setattr(t, k, getattr(self, k))
        return t

    @classmethod
    def merge_augs(cls, result_list, cfg, device=None):
        """Merge the refined tracklets of the test-time augmentations of ONE tracklet
        (LiDARTracklet.merge_augs, lidar_tracklet.py:552-607).  All have the same frames.
        cfg['merge']: 'max' (box of the best-scoring augmentation per frame), 'weighted'
        (score-weighted centre/size, median yaw, mean score) or 'iou_clamped_weighted' (weights of
        augmentations whose box overlaps the first one's by <= cfg['iou_merge_thresh'] are zeroed;
        aligned IoU on the device).  Returns the first tracklet, updated in place."""
        base = result_list[0]
        all_boxes = torch.stack([r.boxes[:, :7] for r in result_list], 0)      # [A, L, 7]
        all_scores = torch.stack([r.scores for r in result_list], 0).clone()   # [A, L]
        num_augs, len_trk = all_scores.shape
        mode = cfg['merge']
        if mode == 'max':
            arg = all_scores.argmax(0)
            cols = torch.arange(len_trk, device=arg.device)
            merged_scores = all_scores[arg, cols]
            merged_boxes = all_boxes[arg, cols]
        elif mode in ('weighted', 'iou_clamped_weighted'):
            if mode == 'iou_clamped_weighted':
                flat = all_boxes.reshape(num_augs * len_trk, 7)
                rep = all_boxes[0].repeat(num_augs, 1)
                if device is not None:
                    flat, rep = flat.to(device), rep.to(device)
                ious = aligned_iou_3d(rep, flat).reshape(num_augs, len_trk).to(all_scores.device)
                ious[0, :] = 1
                all_scores = all_scores * (ious > cfg['iou_merge_thresh']).to(all_scores.dtype)
            w = all_scores[..., None]
            box6 = (all_boxes[..., :6] * w).sum(0) / all_scores.sum(0)[:, None]
            yaw = _median_lower_upper_mean(all_boxes[..., 6])                    # np.median: mean of the middle two
            merged_boxes = torch.cat([box6, yaw[:, None]], 1)
            merged_scores = all_scores.mean(0)
        else:
            raise KeyError(f'unknown TTA merge mode {mode!r}')
        base.boxes = merged_boxes.to(base.boxes.dtype)
        base.scores = merged_scores.to(base.scores.dtype)
        return base

    def concated_boxes_from_ts(self, ts_list):
        """Boxes at the given timestamps (zeros + False where this tracklet has none) (:318-339)."""
        boxes = self.boxes.new_zeros((len(ts_list), 7))
        mask = torch.zeros((len(ts_list),), dtype=torch.bool, device=self.device)
        for i, ts in enumerate(ts_list):
            j = self.ts2index.get(ts, None)
            if j is not None:
                boxes[i] = self.boxes[j, :7]
                mask[i] = True
        return boxes, mask


class AssignResult(object):
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None, gt_inds_host=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels
        self.gt_inds_host = gt_inds_host   # the same indices as a Python list when the assigner built them on the host


class SamplingResult(object):
    """PseudoSampler output (every box kept): positives first, then negatives."""

    def __init__(self, assign_result, bboxes, gt_bboxes):
        host = assign_result.gt_inds_host
        if host is not None:   # (timestamp matching happens on the host: no nonzero(), no read-back)
            self.pos_inds = host_index([i for i, g in enumerate(host) if g > 0], bboxes.device)
            self.neg_inds = host_index([i for i, g in enumerate(host) if g == 0], bboxes.device)
        else:
            self.pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
            self.neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        self.pos_bboxes, self.neg_bboxes = bboxes[self.pos_inds], bboxes[self.neg_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[self.pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = gt_bboxes.new_zeros((0, 7))
        else:
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[self.pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        ready = self.__dict__.get('_bboxes')   # (the batched assignment gathers the rows in this order once for the batch)
        return ready if ready is not None else torch.cat([self.pos_bboxes, self.neg_bboxes])


@BBOX_ASSIGNERS.register_module()
class TrackletAssigner(object):
    """Frame-by-frame assignment by timestamp (tracklet_assigner.py:14-57)."""

    def __init__(self, object_centric=False, iou_thr=0.5):
        self.object_centric, self.iou_thr = object_centric, iou_thr

    def assign(self, trk_pd, trk_gt):
        device = trk_pd.device
        num_gts, num_bboxes = len(trk_gt), len(trk_pd)
        assigned_labels = torch.full((num_bboxes,), -1, dtype=torch.long, device=device)
        scores = trk_pd.concated_scores().detach()
        if num_gts == 0 or num_bboxes == 0:
            fill = 0 if num_gts == 0 else -1
            gt_inds = torch.full((num_bboxes,), fill, dtype=torch.long, device=device)
            res = AssignResult(num_gts, gt_inds, torch.zeros((num_bboxes,), device=device), assigned_labels,
                               gt_inds_host=[fill] * num_bboxes)
            res.scores = scores
            return res
        overlaps = trk_pd.self_ious(trk_gt)
        idx = [trk_gt.get_index_from_ts(ts) + 1 for ts in trk_pd.ts_list]
        if self.object_centric:
            ov = overlaps.tolist()
            idx = [j if ov[i] > self.iou_thr else 0 for i, j in enumerate(idx)]
        gt_inds = host_index(idx, device)
        assigned_labels = torch.where(gt_inds > 0, torch.full_like(assigned_labels, trk_gt.type), assigned_labels)
        res = AssignResult(num_gts, gt_inds, overlaps, assigned_labels, gt_inds_host=idx)
        res.scores = scores
        return res
