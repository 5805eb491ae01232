"""OcOccNet heads -- host mirror of OccAutoEncoder (mmdet3d/models/roi_heads/bbox_heads/
occ_ae_head.py:28-264), OccBBoxHead (ococc_bbox_head.py:37-1309) and the few helpers they
inherit from FullySparseBboxHead (fsd_bbox_head.py:238-272, 442-455, 595-689, 1075-1095).

Same constructor arguments, parameter names (block_list.<i>..., occ_ae_head.point_encoder...,
occ_ae_head.occ_decoder..., trans_enc.layers.<i>..., roi_pos_enc_mlp, conv_cls, conv_reg,
conv_latent, conv_fused) and result dictionaries as the reference, so its configs build
and its checkpoints load.  What differs is underneath:
  * points arrive sorted by RoI from ococc_dynamic_point_pool_mixed, so every scatter is a
    run-length segment reduction (ococc_segment_reduce_f32);
  * every Linear->LN->GELU uses the fused ococc_layernorm_act kernels;
  * the occupancy loss calls the decoder with (RoI features, query points, RoI index) and
    the decoder's first layer is factorised, instead of materialising [R+,K,1536] copies.
"""
import os

import numpy as np
import torch
from torch import nn

from ._lib import const_tensor, log_once
from .bbox import _plain, _rows7, build_bbox_coder, points_box_to_box, rotation_3d_in_axis
from .losses import build_loss, reduce_mean
from .occ import occ_ops
from .occ.layers import PositionalEncoding, SimpleEncoderLayer, TransformerEncoder
from .occ.occ_base import OccDecoder
from .registry import BACKBONES, HEADS
from .sir import SIRLayer, rel_gates
from . import voxel_encoders  # noqa: F401  (registers DynamicVFE / DynamicSimpleVFE)
from .sst.sst_ops import build_mlp, unique_with_inverse


# The temporal transformer's shapes are fixed by the batch layout ([L frames, B tracklets, D]): its forward and backward
# are replayed as two HIP graphs (torch.cuda.make_graphed_callables: static input / output buffers, one graph pair per
# shape) instead of ~60 + ~120 launches issued one by one -- the step at 4 tracklets is bound by the host.  Training
# mode with gradients only; anything else (eval, no_grad, a capture already running, ever-changing shapes) takes the
# eager path.
GRAPH_TRANSFORMER = os.environ.get('OCOCC_GRAPH_TRANSFORMER', '1') == '1'
# Round 3 saw ONE segmentation fault inside hipGraphLaunch with these pairs on: the captured BACKWARD graph is replayed by
# the autograd engine's device thread, i.e. hipGraphLaunch runs on a second host thread while the main thread owns every
# other launch of the process.  With the engine's worker threads off the backward nodes -- and the replay -- run on the
# calling thread, like every other launch (one process per GPU: the worker thread buys nothing).  graphed_call() switches
# them off the first time it makes a pair; OCOCC_GRAPH_AUTOGRAD_THREADS=1 keeps the engine multithreaded
# (tools/soak_graph_pairs.py soaks either configuration in a fresh process).
GRAPH_AUTOGRAD_THREADS = os.environ.get('OCOCC_GRAPH_AUTOGRAD_THREADS', '0') == '1'


class _GraphTable(dict):
    """{shape key: graphed callable} of one owner module, kept ON the module (a graphed callable references its owner:
    in a table keyed weakly by the owner that reference would keep the entry -- graphs, static buffers -- alive for
    ever).  Copies and pickles of the module start with an empty table."""

    def __deepcopy__(self, memo):
        return _GraphTable()

    def __reduce__(self):
        return (_GraphTable, ())


HOST_POSITIVE_COUNTS = True   # False: OccBBoxHead.loss reads its two positive counts back from the device (tests)


class _Graphed(object):
    """`owner in _graphed_encoders` / `_graphed_encoders.get(owner)`: the owner's table, if it has replayed anything"""

    def get(self, owner, default=None):
        t = owner.__dict__.get('_ococc_graphs')
        return t if t else default

    def __contains__(self, owner):
        return bool(owner.__dict__.get('_ococc_graphs'))

    def table(self, owner):
        t = owner.__dict__.get('_ococc_graphs')
        if t is None:
            t = owner.__dict__['_ococc_graphs'] = _GraphTable()
        return t


_graphed_encoders = _Graphed()


class _EncoderCall(nn.Module):
    """positional-argument face of TransformerEncoder.forward for make_graphed_callables (which also needs a Module to
    find the parameters whose gradients the captured backward has to return)"""

    def __init__(self, enc):
        super().__init__()
        self.enc = enc

    def forward(self, feats, pos, mask):
        return self.enc(feats, pos_enc=pos, attn_mask=mask)


def graphed_call(owner, make_wrapper, args, slot=''):
    """``make_wrapper(owner)(*args)`` replayed as a HIP-graph pair when that is possible (training mode, gradients on,
    device tensors of which at least one requires a gradient, every parameter trainable when the graph is made, no
    capture running, at most four shapes seen per owner); None when not -- the caller then runs the eager form."""
    if not (GRAPH_TRANSFORMER and owner.training and torch.is_grad_enabled() and all(a.is_cuda for a in args)
            and any(a.requires_grad for a in args) and not torch.cuda.is_current_stream_capturing()):
        return None
    # ROCm 7.2: with the runtime's graph packet-capture fast path on, replaying a graph after new device allocations
    # faults (graph.GraphedStep refuses to run then; here the eager form runs instead).  The variable has to be in the
    # environment before the HIP runtime loads, i.e. before `import torch`: bench.py and tests/conftest.py set it.
    if os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '') != '0':
        return None
    key = (slot,) + tuple((tuple(a.shape), a.dtype, bool(a.requires_grad)) for a in args)
    table = _graphed_encoders.table(owner)
    g = table.get(key)
    if g is None:
        if len(table) >= 4:   # (a graph pair per shape: not for inputs whose shape keeps changing)
            return None
        sample = tuple((torch.randn_like(a) if a.is_floating_point() else a.clone()).requires_grad_(a.requires_grad)
                       for a in args)
        # (warm-up and capture run on a side stream, and the captured backward keeps the autograd graph of its static
        # outputs: the parameters' AccumulateGrad nodes meet gradients from another stream than the one they were made
        # on, which the engine reports -- the known, intended consequence of graphing a sub-module)
        wrapper = make_wrapper(owner)
        if not all(p.requires_grad for p in wrapper.parameters()):   # (the captured backward returns every parameter's
            table[key] = False                                        # gradient: frozen parameters -> eager, for good)
            return None
        if not GRAPH_AUTOGRAD_THREADS and torch.autograd.is_multithreading_enabled():
            # process-wide and for good: the replayed backward runs inside the CALLER's loss.backward(), there is no scope
            # of ours to restore it in.  Said once, documented in INTEGRATION.md ("Error behaviour"), OCOCC_GRAPH_AUTOGRAD_THREADS=1
            # keeps the engine's threads, OCOCC_GRAPH_TRANSFORMER=0 keeps the eager form.
            torch.autograd.set_multithreading_enabled(False)   # replays on the calling thread from here on (see above)
            log_once('autograd-threads', 'graphed temporal transformer: torch.autograd multithreading switched OFF for this '
                       'process (HIP-graph backward replays run on the calling thread); OCOCC_GRAPH_AUTOGRAD_THREADS=1 keeps it on, '
                       'OCOCC_GRAPH_TRANSFORMER=0 runs the eager form', level='warning')
        quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        was = getattr(torch._C, '_warn_on_accumulate_grad_stream_mismatch', lambda: True)()
        if quiet is not None:
            quiet(False)
        try:
            g = table[key] = torch.cuda.make_graphed_callables(wrapper, sample, allow_unused_input=True)
        finally:
            if quiet is not None:
                quiet(was)   # (silenced for the capture's side stream only; the caller's setting is back afterwards)
    return g(*args) if g is not False else None


def run_encoder(enc, feats, pos, mask):
    if mask is not None and pos is not None and feats.requires_grad:
        out = graphed_call(enc, _EncoderCall, (feats, pos, mask))
        if out is not None:
            return out
    return enc(feats, pos_enc=pos, attn_mask=mask)


class _HeadTail(nn.Module):
    """what OccBBoxHead.forward does behind the transformer (ococc_bbox_head.py:372-400): latent fusion, the fused
    feature and the two prediction MLPs -- fixed shapes [R, .], graphed like the encoder"""

    def __init__(self, head):
        super().__init__()
        self.conv_latent, self.conv_fused, self.conv_cls, self.conv_reg = (head.conv_latent, head.conv_fused, head.conv_cls,
                                                                            head.conv_reg)
        self.fused_mode, self.rcnn_trans = head.fused_mode, head.rcnn_trans

    def forward(self, local_roi_feats, roi_feats_fused, final_cluster_feats):
        if self.fused_mode == 'residual':
            shape_latent = local_roi_feats + self.conv_latent(roi_feats_fused)
        elif self.fused_mode == 'concat':
            shape_latent = self.conv_latent(torch.cat([local_roi_feats, roi_feats_fused], dim=1))
        else:  # concat_residual
            shape_latent = local_roi_feats + self.conv_latent(torch.cat([local_roi_feats, roi_feats_fused], dim=1))
        second = roi_feats_fused if self.rcnn_trans else final_cluster_feats
        fused = self.conv_fused(torch.cat([shape_latent, second], dim=1))
        return shape_latent, self.conv_cls(fused), self.conv_reg(fused)


def _true_rows(mask, count=None):
    """Row numbers where ``mask`` is set.  With ``count`` (the number of set entries, already on the host) the output
    size is known and nothing is read back."""
    if count is not None and hasattr(torch, 'nonzero_static'):
        return torch.nonzero_static(mask, size=int(count)).reshape(-1)
    return torch.nonzero(mask).reshape(-1)


def _filter_rows(pos_batch_idx, filtered_pos_mask, roi_batch_idx, count=None):
    """Row numbers into pos_data for SparseHeadMixin.filter_pos_assigned_but_empty_rois."""
    rb = roi_batch_idx.long()
    pb = pos_batch_idx.long()
    srb, order_r = torch.sort(rb, stable=True)
    spb, order_p = torch.sort(pb, stable=True)
    n = rb.numel()
    # position of every RoI among the RoIs of its sample = sorted position - first sorted position of the sample
    local = torch.arange(n, device=rb.device) - torch.searchsorted(srb, srb)
    sel = _true_rows(filtered_pos_mask[order_r], count)                  # (a read-back of the output size without `count`)
    at = torch.searchsorted(spb, srb[sel]) + local[sel]                    # sorted position inside pos_data
    return order_p[at]


class SparseHeadMixin(object):
    """Helpers OccBBoxHead / OccAutoEncoder take from FullySparseBboxHead."""

    @staticmethod
    def _spare_slot_index(out_coors, num_rois):
        """out_coors with its -1 entries sent to the spare slot ``num_rois`` (int64), kept on the tensor: the mask and the
        aligned features of one encoder call ask for the same index"""
        hit = getattr(out_coors, '_ococc_slot_index', None)
        if hit is None or hit[0] != num_rois or hit[1] != out_coors._version:
            idx = torch.where(out_coors >= 0, out_coors, torch.full_like(out_coors, num_rois)).long()
            hit = out_coors._ococc_slot_index = (num_rois, out_coors._version, idx)
        return hit[2]

    def get_nonempty_roi_mask(self, out_coors, num_rois):
        """fsd_bbox_head.py:238-250."""
        # (-1 entries go to a spare slot instead of being compacted away: no read-back)
        idx = self._spare_slot_index(out_coors, num_rois)
        mask = torch.zeros(num_rois + 1, dtype=torch.bool, device=out_coors.device)
        mask.index_fill_(0, idx, True)
        return mask[:num_rois]

    def align_roi_feature_and_rois(self, features, out_coors, num_rois):
        """Rows of `features` follow the sorted non-empty RoIs; put them at their RoI index,
        zeros for empty RoIs (fsd_bbox_head.py:252-272)."""
        # rows whose coordinate is -1 land in a spare row that is cut off again: no mask compaction, no read-back, and
        # `features` stays connected to the result whether or not any RoI is non-empty
        idx = self._spare_slot_index(out_coors, num_rois)
        new_feature = features.new_zeros((num_rois + 1, features.size(1))).index_copy(0, idx, features)
        return new_feature[:num_rois]

    def filter_pos_assigned_but_empty_rois(self, pos_data, pos_batch_idx, filtered_pos_mask, roi_batch_idx, count=None):
        """fsd_bbox_head.py:442-455: per sample b, the rows of `pos_data` that belong to b, picked at the in-sample
        positions where `filtered_pos_mask` (over the RoIs of b) is set; samples concatenated in order.  The
        reference loops over the samples with three boolean-mask indexings (= three host read-backs) each; here
        the row numbers come from two stable sorts and one `nonzero`, and are reused while the same three index
        tensors are passed again (the loss filters four tensors with them)."""
        key = (pos_batch_idx, filtered_pos_mask, roi_batch_idx)
        cached = getattr(self, '_filter_rows_cache', None)
        if cached is not None and all(a is b for a, b in zip(cached[0], key)) \
                and cached[1] == tuple(t._version for t in key):
            rows = cached[2]
        else:
            rows = _filter_rows(*key, count=count)
            self._filter_rows_cache = (key, tuple(t._version for t in key), rows)
        return pos_data[rows]

    def get_class_wise_box_weights(self, weights, gt_labels, cfg):
        class_wise_weight = cfg.get('class_wise_box_weights', None)
        if class_wise_weight is None:
            return weights
        all_gt = torch.cat([gt_labels, gt_labels.new_full((len(weights) - len(gt_labels),), -1)], 0)
        for i in range(self.num_classes):
            weights[all_gt == i] *= class_wise_weight[i]
        return weights

    def get_multi_class_soft_label(self, ious, pos_gt_labels, cfg):
        """IoU-interpolated soft labels between cls_neg_thr and cls_pos_thr
        (fsd_bbox_head.py:627-689)."""
        pos_thrs, neg_thrs = cfg['cls_pos_thr'], cfg['cls_neg_thr']
        if isinstance(pos_thrs, float):
            pos_thrs, neg_thrs = [pos_thrs] * self.num_classes, [neg_thrs] * self.num_classes
        num_samples, num_pos = ious.size(0), pos_gt_labels.size(0)
        all_gt = torch.cat([pos_gt_labels, pos_gt_labels.new_full((num_samples - num_pos,), -1)], 0)
        all_label = ious.new_zeros(num_samples)
        for i in range(self.num_classes):
            # (elementwise selects instead of the reference's boolean-mask assignments: same values, no read-back)
            pos = ious > pos_thrs[i]
            interval = (~pos) & ~(ious < neg_thrs[i])
            lab = torch.where(interval, (ious - neg_thrs[i]) / (pos_thrs[i] - neg_thrs[i]), pos.to(ious.dtype))
            all_label = torch.where(all_gt == i, lab, all_label)
        label_weights = (all_label >= 0).float()
        cw = cfg.get('class_wise_cls_weights', None)
        if cw is not None:
            for i in range(self.num_classes):
                label_weights[all_gt == i] *= cw[i]
        return all_label, label_weights

    def decode_from_rois(self, rois, bbox_pred):
        """fsd_bbox_head.py:1075-1095: canonical deltas -> boxes in the ego frame."""
        roi_boxes = rois[..., 1:]
        roi_ry = roi_boxes[..., 6].view(-1)
        roi_xyz = roi_boxes[..., 0:3].view(-1, 3)
        local = roi_boxes.clone().detach()
        local[..., 0:3] = 0
        if local.size(1) == 9:
            bbox_pred = torch.nn.functional.pad(bbox_pred, (0, 2), 'constant', 0)
        boxes = self.bbox_coder.decode(local, bbox_pred)
        boxes[..., 0:3] = rotation_3d_in_axis(boxes[..., 0:3].unsqueeze(1), roi_ry + np.pi / 2, axis=2).squeeze(1)
        boxes[:, 0:3] += roi_xyz
        return boxes


    def get_bboxes_from_tracklet(self, rois, cls_score, bbox_pred, valid_roi_mask, class_labels, class_pred, img_metas,
                                 gt_rois=None, cfg=None):
        """fsd_bbox_head.py:994-1073: per sample (boxes [n,7], sigmoid scores, labels, non-empty mask); empty RoIs are
        kept and flagged, the caller decides.  test_cfg switches: identical_decode (return the proposals and their
        scores), replace_center / replace_size / replace_yaw (oracle experiments with the matched GT boxes)."""
        assert rois.size(0) == cls_score.size(0) == bbox_pred.size(0)
        tc = self.test_cfg or {}
        scores = cls_score.sigmoid()
        if tc.get('identical_decode', False):
            boxes, scores = rois[:, 1:], class_pred
        else:
            boxes = self.decode_from_rois(rois, bbox_pred)
        for key, dst, src in (('replace_center', slice(0, 3), slice(1, 4)), ('replace_size', slice(3, 6), slice(4, 7)),
                              ('replace_yaw', slice(6, 7), slice(7, 8))):
            if tc.get(key, False):
                m = gt_rois[:, 0].bool()
                boxes[m, dst] = gt_rois[m, src]
        batch = rois[..., 0]
        out = []
        for b in range(int(batch.max().item() + 1)):
            m = batch == b
            out.append((boxes[m], scores[m].view(-1), class_labels[m], valid_roi_mask[m]))
        return out


@HEADS.register_module()
class OccAutoEncoder(nn.Module, SparseHeadMixin):
    """Point encoder (SIR) + implicit occupancy decoder (occ_ae_head.py:28-264)."""

    def __init__(self, backbone, occ_decoder, voxel_size,
                 loss_occ_ae=dict(type='CrossEntropyLoss', reduction='none', use_sigmoid=True, loss_weight=1.0),
                 scale_wlh=[1.0, 1.0, 1.0], offset_wlh=[0.0, 0.0, 0.0], online_sample_size=-1,
                 balance_sample=False, with_voxelize_centers=False, compensate_encoder_coors=False,
                 add_train_prob=0.0, init_cfg=None, train_cfg=None, test_cfg=None):
        super().__init__()
        self.point_encoder = BACKBONES.build(backbone)
        self.occ_decoder = OccDecoder(**occ_decoder)
        self.loss_occ_ae = build_loss(loss_occ_ae)
        self.voxel_size = voxel_size
        self.scale_wlh, self.offset_wlh = scale_wlh, offset_wlh
        self.online_sample_size, self.balance_sample = online_sample_size, balance_sample
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.with_voxelize_centers = with_voxelize_centers
        self.compensate_encoder_coors = compensate_encoder_coors
        self.add_train_prob = add_train_prob
        self.loss_need_squeeze = loss_occ_ae['type'] == 'CrossEntropyLoss' and loss_occ_ae['use_sigmoid']

    # ------------------------------------------------------------------ observation rasterisation
    def sample_observation(self, local_xyz, rois, pts_roi_inds, downsample_size=-1, balance_sample=False):
        """occ_ae_head.py:65-201: rasterise the pooled points of every RoI into its dense 0.2 m grid (label 1 =
        a point fell into the cell) and return (cell centres [S,3], labels [S], RoI index [S]), optionally
        sub-sampled per RoI.

        The reference builds one [X,Y,Z] label volume per RoI in a Python loop; here all grids are one flat
        list of cells (occ_ops.dense_voxel_centers_batched) and the labels are ONE scatter of the points'
        flat cell ids -- the "point-to-voxel scatter over per-object occupancy grids".  Without sub-sampling
        the result is identical (same cells, same order); the sampled variants draw from the same
        distributions with torch's device RNG instead of per-RoI torch.multinomial calls."""
        assert rois.size(1) in (8, 10)
        if not self.compensate_encoder_coors:
            r = const_tensor((np.pi / 2,), local_xyz.device, local_xyz.dtype)
            local_xyz = rotation_3d_in_axis(local_xyz[None, :, :], r, axis=2).squeeze(0)
        R = rois.size(0)
        dev = local_xyz.device
        if R == 0:
            return local_xyz.new_zeros((0, 3)), local_xyz.new_zeros((0,)), local_xyz.new_zeros((0,))
        pts_roi_inds = pts_roi_inds.long()
        coors = occ_ops.quantize_points(local_xyz, rois, pts_roi_inds, self.voxel_size, scale_wlh=self.scale_wlh,
                                        offset_wlh=self.offset_wlh)                                  # [M,3] long
        centers, box, k = occ_ops.dense_voxel_centers_batched(rois[:, 4:7], self.voxel_size, self.scale_wlh,
                                                              self.offset_wlh)
        size = rois[:, 4:7] * rois.new_tensor(self.scale_wlh) + rois.new_tensor(self.offset_wlh)
        dims = torch.ceil(size / self.voxel_size).to(torch.long)
        start = torch.cumsum(k, 0) - k
        d = dims[pts_roi_inds]
        valid = ((coors < d) & (coors >= 0)).all(dim=1)      # points exactly on the far boundary fall outside
        flat = start[pts_roi_inds] + (coors[:, 0] * d[:, 1] + coors[:, 1]) * d[:, 2] + coors[:, 2]
        labels = torch.zeros(centers.size(0), dtype=torch.long, device=dev)
        labels[flat[valid]] = 1
        keep = None
        if balance_sample:
            keep = self._balanced_subset(labels, box, k, start, downsample_size)
        elif downsample_size > 0:
            keep = self._weighted_subset(labels, box, k, start, downsample_size)
        if keep is not None:
            centers, labels, box = centers[keep], labels[keep], box[keep]
        return centers, labels, box

    @staticmethod
    def _rank_in_roi(key, box, start):
        """rank of every cell among the cells of its RoI when ordered by descending key"""
        order = torch.argsort(box.to(torch.float64) * 4.0 - key.to(torch.float64).clamp(0, 1) , stable=True)
        rank = torch.empty_like(order)
        rank[order] = torch.arange(order.numel(), device=order.device)
        return rank - start[box]

    def _weighted_subset(self, labels, box, k, start, n):
        """RoIs with more than n cells keep n of them, drawn without replacement with weight 100 for observed
        cells and 1 for the rest (occ_ae_head.py:167-181; Efraimidis-Spirakis keys u^(1/w) give the
        distribution of sequential weighted draws)."""
        w = torch.where(labels == 1, 100.0, 1.0)
        key = torch.rand(labels.numel(), device=labels.device, dtype=torch.float64).clamp_min(1e-300) ** (1.0 / w)
        rank = self._rank_in_roi(key, box, start)
        keep = (rank < n) | (k[box] <= n)
        return torch.nonzero(keep).squeeze(1)

    def _balanced_subset(self, labels, box, k, start, n):
        """occ_ae_head.py:129-166: all observed cells of a RoI plus as many free cells (with replacement when
        there are fewer free than observed cells); a RoI without observed cells contributes its first cell;
        then at most n cells per RoI, uniformly."""
        dev = labels.device
        R = k.numel()
        pos = labels == 1
        npos = torch.zeros(R, dtype=torch.long, device=dev).index_add_(0, box, pos.long())
        nneg = k - npos
        u = torch.rand(labels.numel(), device=dev, dtype=torch.float64)
        # free cells ranked by a random key inside their RoI (observed cells pushed to the end)
        rank_neg = self._rank_in_roi(torch.where(pos, torch.zeros_like(u), 0.5 + 0.5 * u), box, start)
        take_neg = (~pos) & (rank_neg < npos[box]) & (npos[box] <= nneg[box])
        first_only = (npos[box] == 0) & (torch.arange(labels.numel(), device=dev) == start[box])
        # (a RoI whose cells are ALL observed is skipped, as the reference's `continue` does)
        keep = torch.nonzero((pos & (nneg[box] > 0)) | take_neg | first_only).squeeze(1)
        # RoIs with fewer free than observed cells draw their free cells WITH replacement
        short = torch.nonzero((npos > nneg) & (nneg > 0)).squeeze(1)
        if short.numel() > 0:
            neg_idx = torch.nonzero(~pos).squeeze(1)
            neg_start = torch.cumsum(nneg, 0) - nneg
            reps = npos[short]
            rb = torch.repeat_interleave(short, reps)
            pick = torch.minimum((torch.rand(rb.numel(), device=dev) * nneg[rb]).long(), nneg[rb] - 1)
            keep = torch.cat([keep, neg_idx[neg_start[rb] + pick]])
            keep = keep[torch.argsort(box[keep], stable=True)]
        if n > 0:
            kb = box[keep]
            cnt = torch.zeros(R, dtype=torch.long, device=dev).index_add_(0, kb, torch.ones_like(kb))
            st = torch.cumsum(cnt, 0) - cnt
            r2 = self._rank_in_roi(torch.rand(keep.numel(), device=dev, dtype=torch.float64), kb, st)
            keep = keep[(r2 < n) | (cnt[kb] <= n)]
        return keep

    # ------------------------------------------------------------------ auto-encoder stage (occ_ae_head.py:270-344)
    def forward_train_ae(self, pts_xyz, pts_features, pts_info, roi_inds, rois, start_add_train=False):
        local_roi_feats, nonempty_roi_mask, local_xyz = self.encode(pts_xyz, pts_features, pts_info, roi_inds, rois)
        if start_add_train and torch.rand(1).item() < self.add_train_prob:
            # merge every RoI with a random partner: max of the two features, union of the two point sets,
            # the larger of the two boxes (occ_ae_head.py:277-318).  perm[i] = partner of RoI i; the points of
            # RoI j are appended to RoI inv[j], no per-RoI loop
            perm = torch.randperm(len(rois), device=rois.device)
            local_roi_feats = torch.max(torch.stack([local_roi_feats, local_roi_feats[perm]], dim=0), dim=0)[0]
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(len(rois), device=rois.device)
            new_xyz = torch.cat([local_xyz, local_xyz], dim=0)
            new_inds = torch.cat([roi_inds.long(), inv[roi_inds.long()]], dim=0)
            new_rois = rois.clone()
            new_rois[:, 4:7] = torch.max(torch.stack([rois[:, 4:7], rois[perm][:, 4:7]], dim=0), dim=0)[0]
            smp_xyz, labels, smp_inds = self._sample_obs(new_xyz, new_rois, new_inds)
        else:
            smp_xyz, labels, smp_inds = self._sample_obs(local_xyz, rois, roi_inds)
        occ_preds = self.decode(local_roi_feats, smp_xyz, smp_inds)
        return self.loss(occ_preds, local_roi_feats, smp_xyz, smp_inds, labels, nonempty_roi_mask)

    def _sample_obs(self, local_xyz, rois, roi_inds):
        return self.sample_observation(local_xyz, rois, roi_inds, downsample_size=self.online_sample_size,
                                       balance_sample=self.balance_sample)

    def loss(self, occ_preds, local_roi_feats, smp_pts_xyz_local, smp_pts_roi_inds, obs_occ_labels, nonempty_roi_mask):
        """occ_ae_head.py:451-509."""
        num_occupied = obs_occ_labels.sum().float()
        num_free = obs_occ_labels.numel() - num_occupied
        per_points_masks = nonempty_roi_mask[smp_pts_roi_inds]
        num_valid_occupied = ((obs_occ_labels == 1) & (per_points_masks == 1)).sum().float()
        num_valid_free = ((obs_occ_labels == 0) & (per_points_masks == 1)).sum().float()
        occ_preds = occ_preds.view(-1) if self.loss_need_squeeze else occ_preds.view(-1, 1)
        assert len(smp_pts_xyz_local) > 0
        loss_ae = self.loss_occ_ae(occ_preds, obs_occ_labels.view(-1)).mean()
        assert (smp_pts_roi_inds >= 0).all()
        if self.loss_need_squeeze:
            pred_cls = (occ_preds.sigmoid() > 0.5).long().view(-1)
        else:
            pred_cls = (occ_preds.sigmoid() < 0.5).long().view(-1)
        num_pred_occupied = pred_cls.sum().float()
        num_pred_free = pred_cls.numel() - num_pred_occupied
        num_gt_occupied = obs_occ_labels.sum().float()
        num_gt_free = obs_occ_labels.numel() - num_gt_occupied
        num_correct_occupied = ((pred_cls == 1) & (obs_occ_labels == 1)).sum().float()
        num_correct_free = ((pred_cls == 0) & (obs_occ_labels == 0)).sum().float()
        return dict(recall_free=num_correct_free / (num_gt_free + 1e-6),
                    recall_occupied=num_correct_occupied / (num_gt_occupied + 1e-6),
                    precision_free=num_correct_free / (num_pred_free + 1e-6),
                    precision_occupied=num_correct_occupied / (num_pred_occupied + 1e-6),
                    loss_ae=loss_ae, num_occupied=num_occupied, num_free=num_free,
                    num_valid_occupied=num_valid_occupied, num_valid_free=num_valid_free)

    def online_tuning_forward(self, roi_features, pts_smp, pts_labels, pts_weights, pts_roi_inds, num_ttt_iter,
                              apply_jitter=False):
        """occ_ae_head.py:346-391: test-time tuning of the RoI embeddings -- num_ttt_iter Adam(lr 0.01) steps on
        the observation loss with the decoder frozen."""
        roi_embed = roi_features.clone().detach()
        roi_embed.requires_grad = True
        if pts_weights is None:
            pts_weights = torch.ones_like(pts_labels, dtype=torch.float32)
        train_state = self.training
        with torch.enable_grad():
            optimizer = torch.optim.Adam([roi_embed], lr=0.01)
            scheduler = torch.optim.lr_scheduler.StepLR(optimizer, 1000, 0.1)
            self.eval()
            for param in self.parameters():
                param.requires_grad = False
            for _ in range(num_ttt_iter):
                optimizer.zero_grad()
                occ_preds = self.decode(roi_embed, pts_smp, pts_roi_inds)
                occ_preds = occ_preds.view(-1) if self.loss_need_squeeze else occ_preds.view(-1, 1)
                if len(occ_preds) > 0:
                    loss_ae = self.loss_occ_ae(occ_preds, pts_labels.view(-1), pts_weights).mean()
                    loss_ae.backward()
                    optimizer.step()
                    scheduler.step()
        self.train(train_state)
        for param in self.parameters():
            param.requires_grad = train_state
        return roi_embed

    def forward_test_ae(self, pts_xyz, pts_features, pts_info, roi_inds, rois):
        """occ_ae_head.py:393-419."""
        local_roi_feats, _, local_xyz = self.encode(pts_xyz, pts_features, pts_info, roi_inds, rois)
        if self.test_cfg and self.test_cfg.get('online_tuning', False):
            smp_xyz, labels, smp_inds = self.sample_observation(
                local_xyz, rois, roi_inds, downsample_size=self.test_cfg.get('downsample_size', -1),
                balance_sample=self.test_cfg.get('balance_sample', False))
            local_roi_feats = self.online_tuning_forward(local_roi_feats, smp_xyz, labels, None, smp_inds,
                                                         self.test_cfg.get('num_iter', 10))
        return self.get_occ(local_roi_feats, rois)

    def encode(self, pts_xyz, pts_features, pts_info, roi_inds, rois, point_encoder=None, cat_global_xyz=False):
        """occ_ae_head.py:203-264 -> (local_roi_feats [R,D], nonempty mask [R], rotated local xyz)."""
        local_xyz = pts_info['local_xyz']
        if self.compensate_encoder_coors:  # the pi/2 frame fix the pooling op leaves to callers
            r = const_tensor((np.pi / 2,), local_xyz.device, local_xyz.dtype)
            local_xyz = rotation_3d_in_axis(local_xyz[None, :, :], r, axis=2).squeeze(0)
        boundary_offset, is_in_margin = pts_info['boundary_offset'], pts_info['is_in_margin']
        parts = [boundary_offset, is_in_margin[:, None]]
        if pts_features is not None:
            parts = [pts_features] + parts
        if cat_global_xyz:
            parts.append(pts_xyz)
        if self.with_voxelize_centers:
            assert self.compensate_encoder_coors
            parts.append(occ_ops.quantize_points(local_xyz, rois, roi_inds, self.voxel_size, self.scale_wlh,
                                                 self.offset_wlh, to_center=True))
        out_feats = torch.cat(parts, 1)
        if point_encoder is None:
            point_encoder = self.point_encoder
        out_feats, final_cluster_feats, out_coors = point_encoder(local_xyz, out_feats, roi_inds, dims=[len(rois)])
        nonempty_roi_mask = self.get_nonempty_roi_mask(out_coors, len(rois))
        final_cluster_feats = self.align_roi_feature_and_rois(final_cluster_feats, out_coors, len(rois))
        return final_cluster_feats, nonempty_roi_mask, local_xyz

    def decode(self, roi_feats, smp_pts_xyz_local, smp_pts_roi_inds):
        return self.occ_decoder(roi_feats, smp_pts_xyz_local, smp_pts_roi_inds)

    def get_occ(self, local_roi_feats, rois, transform=True):
        """occ_ae_head.py:421-438: explicit occupancy of every RoI (dense grid decode)."""
        return self.occ_decoder.get_occ(local_roi_feats, rois, self.voxel_size, self.scale_wlh, self.offset_wlh,
                                        transform=transform)

    def get_roi_occ(self, local_roi_feats, rois, transform=True, return_score=False):
        """occ_ae_head.py:440-449."""
        return self.occ_decoder.get_roi_occ(local_roi_feats, rois, self.voxel_size, self.scale_wlh,
                                            self.offset_wlh, transform, return_score)


class _Sampling(object):
    """What get_targets reads from an mmdet SamplingResult (tracklet_roi_head_occ.py:880-991)."""

    def __init__(self, pos_bboxes, pos_gt_bboxes, iou, pos_gt_labels, occ_labels, occ_scores):
        self.pos_bboxes, self.pos_gt_bboxes, self.iou = pos_bboxes, pos_gt_bboxes, iou
        self.pos_gt_labels, self.occ_labels, self.occ_scores = pos_gt_labels, occ_labels, occ_scores


@HEADS.register_module()
class OccBBoxHead(nn.Module, SparseHeadMixin):

    def __init__(self, num_blocks, in_channels, feat_channels, rel_mlp_hidden_dims, rel_mlp_in_channels,
                 with_rel_mlp=True, with_cluster_center=False, with_distance=False, mode='max',
                 xyz_normalizer=[20, 20, 4], geo_input=True, dropout=0, unique_once=True, occ_ae_head=None,
                 roi_feature_channels=None, init_cfg=None, debug=False, fixed_ae=True, attn_num_head=4,
                 attn_ffn_dim=2048, attn_dropout=0.1,
                 loss_occ_comp=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='none', loss_weight=1.0),
                 num_classes=1, bbox_coder=dict(type='DeltaXYZWLHRBBoxCoder'), occ_label_thresh=0.8,
                 reg_mlp=None, cls_mlp=None, latent_mlp=None, fusion_mlp=None, act='gelu',
                 norm_cfg=dict(type='LN', eps=1e-3),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=2.0),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='none', loss_weight=1.0),
                 cls_dropout=0, reg_dropout=0, latent_dropout=0, fusion_dropout=0, with_corner_loss=False,
                 with_roi_pos_encoding=False, roi_pos_enc_mlp=None, roi_enc_dropout=0, num_enc_layers=1,
                 fused_mode='residual', rcnn_trans=True, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.box_code_size = self.bbox_coder.code_size
        self.occ_ae_head = HEADS.build(occ_ae_head)
        self.debug, self.fixed_ae = debug, fixed_ae
        if self.fixed_ae:
            for p in self.occ_ae_head.parameters():
                p.requires_grad = False
            self.occ_ae_head.eval()
        self.with_corner_loss = with_corner_loss
        self.trans_enc = TransformerEncoder(
            SimpleEncoderLayer(roi_feature_channels, attn_num_head, dim_feedforward=attn_ffn_dim,
                               dropout=attn_dropout), num_enc_layers)
        self.pos_enc = PositionalEncoding(roi_feature_channels)
        self.loss_occ_comp = build_loss(loss_occ_comp)
        self.num_classes = num_classes
        self.occ_label_thresh = occ_label_thresh
        self.with_roi_pos_encoding = with_roi_pos_encoding
        self.roi_feature_channels = D = roi_feature_channels
        if with_roi_pos_encoding:
            self.roi_pos_enc_mlp = build_mlp(7, list(roi_pos_enc_mlp) + [D], norm_cfg, True, act=act,
                                             dropout=roi_enc_dropout)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.conv_cls = build_mlp(D, list(cls_mlp) + [1], norm_cfg, True, act=act, dropout=cls_dropout) \
            if cls_mlp is not None else nn.Linear(D, 1)
        self.conv_reg = build_mlp(D, list(reg_mlp) + [self.box_code_size], norm_cfg, True, act=act,
                                  dropout=reg_dropout) if reg_mlp is not None else nn.Linear(D, self.box_code_size)
        self.fused_mode = fused_mode
        if fused_mode == 'residual':
            latent_in = D
        elif fused_mode in ('concat', 'concat_residual'):
            latent_in = D * 2
        else:
            raise NotImplementedError(f'Unknown fused_mode: {fused_mode}')
        self.conv_latent = build_mlp(latent_in, list(latent_mlp) + [D], norm_cfg, True, act=act,
                                     dropout=latent_dropout) if latent_mlp is not None else nn.Linear(latent_in, D)
        self.conv_fused = build_mlp(D * 2, list(fusion_mlp) + [D], norm_cfg, True, act=act,
                                    dropout=fusion_dropout) if fusion_mlp is not None else nn.Linear(D * 2, D)
        self.geo_input, self.unique_once, self.num_blocks = geo_input, unique_once, num_blocks
        self.block_list = nn.ModuleList([
            SIRLayer(in_channels=in_channels[i], feat_channels=feat_channels[i], with_distance=with_distance,
                     with_cluster_center=with_cluster_center, with_rel_mlp=with_rel_mlp,
                     rel_mlp_hidden_dims=rel_mlp_hidden_dims[i], rel_mlp_in_channel=rel_mlp_in_channels[i],
                     with_voxel_center=False, voxel_size=[0.1, 0.1, 0.1],
                     point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4], norm_cfg=norm_cfg, mode=mode,
                     fusion_layer=None, return_point_feats=i != num_blocks - 1, return_inv=False,
                     rel_dist_scaler=10.0, xyz_normalizer=xyz_normalizer, act=act, dropout=dropout)
            for i in range(num_blocks)])
        self.rcnn_trans = rcnn_trans
        self.train_cfg = train_cfg if train_cfg is not None else {}
        self.test_cfg = test_cfg if test_cfg is not None else {}

    # ------------------------------------------------------------------ forward
    def roi_encode(self, pts_xyz, pts_features, pts_info, roi_inds, rois):
        """ococc_bbox_head.py:237-316: 6 SIRLayers over the pooled points of each RoI."""
        assert pts_features.size(0) > 0
        rois = rois[:, 1:]
        rel_xyz = pts_xyz[:, :3] - rois[:, :3][roi_inds.long()]
        if self.unique_once:
            new_coors, unq_inv = unique_with_inverse(roi_inds, [len(rois)])
        else:
            new_coors = unq_inv = None
        out_feats = pts_features
        f_cluster = torch.cat([pts_info['local_xyz'], pts_info['boundary_offset'],
                               pts_info['is_in_margin'][:, None], rel_xyz], dim=-1)
        cluster_feat_list = []
        geo = f_cluster / 10 if self.geo_input else None   # (the same for every block: once, and one concatenation per block)
        # the rel_mlp gates of all blocks from the offsets they share, in one launch (None: every block runs its own)
        gates = rel_gates(self.block_list, f_cluster) if self.unique_once else None
        for i, block in enumerate(self.block_list):
            in_feats = torch.cat([pts_xyz, out_feats] if geo is None else [pts_xyz, out_feats, geo], 1)
            kw = {} if gates is None else {'gate': gates[i]}
            if i < self.num_blocks - 1:
                out_feats, out_cluster_feats = block(in_feats, roi_inds, f_cluster, unq_inv_once=unq_inv,
                                                     new_coors_once=new_coors, **kw)
            else:
                out_cluster_feats, out_coors = block(in_feats, roi_inds, f_cluster, unq_inv_once=unq_inv,
                                                     new_coors_once=new_coors, **kw)
            cluster_feat_list.append(out_cluster_feats)
        final_cluster_feats = torch.cat(cluster_feat_list, dim=1)
        nonempty_roi_mask = self.get_nonempty_roi_mask(out_coors, len(rois))
        counts = getattr(roi_inds, '_ococc_roi_counts', None)   # (point_pool: points per RoI, already on the host)
        if counts is not None and len(counts) == len(rois):
            nonempty_roi_mask._ococc_host = [c > 0 for c in counts]
        final_cluster_feats = self.align_roi_feature_and_rois(final_cluster_feats, out_coors, len(rois))
        return final_cluster_feats, nonempty_roi_mask, out_coors

    def forward(self, pts_xyz, pts_features, pts_info, roi_inds, rois, roi_frame_inds):
        """ococc_bbox_head.py:319-400 -> dict(fused_roi_feats, nonempty_roi_mask, ori_roi_feats,
        cls_score, bbox_pred)."""
        if pts_xyz.size(0) == 0:
            final_cluster_feats = pts_features.new_zeros((len(rois), self.roi_feature_channels))
            nonempty_roi_mask = pts_features.new_zeros(len(rois), dtype=torch.bool)
        else:
            final_cluster_feats, nonempty_roi_mask, _ = self.roi_encode(pts_xyz, pts_features, pts_info,
                                                                        roi_inds, rois)
        local_roi_feats, _, local_xyz = self.occ_ae_head.encode(pts_xyz, pts_features[:, :2], pts_info,
                                                                roi_inds, rois)
        roi_feats_fused = self.transformer_forward(rois, roi_frame_inds, final_cluster_feats, nonempty_roi_mask)
        tail = None
        if local_roi_feats.requires_grad and roi_feats_fused.requires_grad and final_cluster_feats.requires_grad:
            tail = graphed_call(self, _HeadTail, (local_roi_feats, roi_feats_fused, final_cluster_feats), slot='tail')
        if tail is not None:
            shape_latent, cls_score, bbox_pred = tail
            return dict(fused_roi_feats=shape_latent, nonempty_roi_mask=nonempty_roi_mask, ori_roi_feats=local_roi_feats,
                        cls_score=cls_score, bbox_pred=bbox_pred)
        if self.fused_mode == 'residual':
            shape_latent = local_roi_feats + self.conv_latent(roi_feats_fused)
        elif self.fused_mode == 'concat':
            shape_latent = self.conv_latent(torch.cat([local_roi_feats, roi_feats_fused], dim=1))
        else:  # concat_residual
            shape_latent = local_roi_feats + self.conv_latent(torch.cat([local_roi_feats, roi_feats_fused], dim=1))
        ret = dict(fused_roi_feats=shape_latent, nonempty_roi_mask=nonempty_roi_mask, ori_roi_feats=local_roi_feats)
        second = roi_feats_fused if self.rcnn_trans else final_cluster_feats
        fused = self.conv_fused(torch.cat([shape_latent, second], dim=1))
        ret.update(cls_score=self.conv_cls(fused), bbox_pred=self.conv_reg(fused))
        return ret

    # ------------------------------------------------------------------ temporal transformer
    def get_occ(self, local_roi_feats, rois, transform=True, ori_roi_feats=None):
        """ococc_bbox_head.py:813-841: explicit occupancy of every RoI from the fused features (and, when
        given, the union with the decode of the single-frame features)."""
        occ_list = self.occ_ae_head.get_occ(local_roi_feats, rois, transform=transform)
        if ori_roi_feats is None:
            return occ_list
        ori_list = self.occ_ae_head.get_occ(ori_roi_feats, rois, transform=transform)
        return [[torch.cat([a, b], dim=0) for a, b in zip(occs, oris)] for occs, oris in zip(occ_list, ori_list)]

    def transformer_forward(self, rois, roi_frame_inds, roi_feats, nonempty_roi_mask, trans_enc=None):
        if not self.training or self.train_cfg.get('fixed_length', True):
            return self.transformer_forward_fixed_length(rois, roi_frame_inds, roi_feats, nonempty_roi_mask, trans_enc)
        return self.transformer_forward_various_length(rois, roi_frame_inds, roi_feats, nonempty_roi_mask, trans_enc)

    def reorder_feats(self, feats, roi_frame_inds, roi_batch_inds, sort_batch_indices=None,
                      sort_frame_indices=None, batch_size=None):
        """Rows -> [B, L, C] sorted by (batch, frame) (ococc_bbox_head.py:997-1019).  ``batch_size``: the caller's B when
        it has read it back already (one read-back per forward instead of one per call)."""
        B = (batch_size if batch_size is not None else sort_frame_indices.shape[0] if sort_frame_indices is not None
             else int(roi_batch_inds.max().item() + 1))
        L = roi_frame_inds.numel() // B
        if sort_batch_indices is None or sort_frame_indices is None:
            sort_batch_indices = torch.argsort(roi_batch_inds)
            sort_frame_indices = torch.argsort(roi_frame_inds[sort_batch_indices].view(B, L), dim=1)
        feats = feats[sort_batch_indices].view(B, L, -1)
        feats = torch.gather(feats, 1, sort_frame_indices[:, :, None].expand(-1, -1, feats.shape[-1]))
        return feats, sort_batch_indices, sort_frame_indices

    def inverse_reorder_feats(self, feats, sort_batch_indices, sort_frame_indices):
        B, L = sort_frame_indices.shape
        inv = torch.argsort(sort_frame_indices.view(B, L), dim=1)
        feats = feats.view(B, L, -1)
        feats = torch.gather(feats, 1, inv[:, :, None].expand(-1, -1, feats.shape[-1]))
        return feats.view(-1, feats.shape[-1])[sort_batch_indices.argsort()]

    def get_future_mask(self, L, device, window_size=-1):
        """True = may not attend: strictly future frames, optionally frames older than the window."""
        if not self.training:
            window_size = self.test_cfg.get('attn_window_size', -1)
        mask = torch.triu(torch.ones(L, L, dtype=torch.bool, device=device), diagonal=1)
        if window_size > 0:
            for i in range(window_size - 1, L):
                mask[i, :i - window_size + 1] = 1
        return mask

    def transformer_forward_fixed_length(self, rois, roi_frame_inds, roi_feats, nonempty_roi_mask, trans_enc=None):
        """ococc_bbox_head.py:849-908."""
        rois_batch_idx = rois[:, 0]
        B = getattr(rois, '_ococc_batch_size', None)   # (set by bbox3d2roi from the list's shapes: no read-back)
        if not B:
            B = int(rois_batch_idx.max().item() + 1)
        L = roi_frame_inds.numel() // B
        assert L * B == roi_frame_inds.numel()
        re_feats, sb, sf = self.reorder_feats(roi_feats, roi_frame_inds, rois_batch_idx, batch_size=B)
        re_frames = self.reorder_feats(roi_frame_inds.clone(), roi_frame_inds, rois_batch_idx, sb, sf)[0].squeeze(-1)
        re_feats = re_feats.view(B, L, re_feats.shape[-1]).permute(1, 0, 2)  # [L, B, D]
        pos_embed = self.pos_enc(re_frames.transpose(0, 1))
        if self.with_roi_pos_encoding:
            re_rois = self.reorder_feats(rois[:, 1:], roi_frame_inds, rois_batch_idx, sb, sf)[0]
            pos_embed = pos_embed + self.roi_pos_enc_mlp(re_rois).transpose(0, 1)
        if not self.training and self.test_cfg.get('allow_attn_future', False):
            future_mask = None
        else:
            future_mask = self.get_future_mask(L, re_feats.device)
        enc = self.trans_enc if trans_enc is None else trans_enc
        out = run_encoder(enc, re_feats, pos_embed, future_mask).transpose(0, 1)
        return self.inverse_reorder_feats(out, sb, sf)

    def transformer_forward_various_length(self, rois, roi_frame_inds, roi_feats, nonempty_roi_mask, trans_enc=None):
        """Tracklets of unequal length: pad to the longest, key-padding mask (ococc_bbox_head.py:911-995)."""
        rois_batch_idx = rois[:, 0]
        B = int(rois_batch_idx.max().item() + 1)
        feats_l, rois_l, rev_l = [], [], []
        for b in range(B):
            m = rois_batch_idx == b
            order = roi_frame_inds[m].argsort()
            rev_l.append(order.argsort())
            feats_l.append(roi_feats[m][order])
            if self.with_roi_pos_encoding:
                rois_l.append(rois[m][order][:, 1:])
        max_len = max(len(x) for x in feats_l)
        pad = lambda t: torch.nn.functional.pad(t, (0, 0, 0, max_len - len(t)), 'constant', 0)
        feats = torch.stack([pad(f) for f in feats_l], 0)  # [B, max_len, D]
        key_padding = torch.stack([torch.arange(max_len, device=feats.device) >= len(f) for f in feats_l], 0)
        frame_inds = torch.arange(max_len, device=feats.device)[None, :].repeat(len(feats_l), 1)
        pos_embed = self.pos_enc(frame_inds.transpose(0, 1))
        if self.with_roi_pos_encoding:
            pos_embed = pos_embed + self.roi_pos_enc_mlp(torch.stack([pad(r) for r in rois_l], 0)).transpose(0, 1)
        enc = self.trans_enc if trans_enc is None else trans_enc
        out = enc(feats.permute(1, 0, 2), pos_enc=pos_embed, key_padding_mask=key_padding,
                  attn_mask=self.get_future_mask(max_len, feats.device)).transpose(0, 1)
        return torch.cat([out[i][:len(feats_l[i])][rev_l[i]] for i in range(len(feats_l))], 0)

    # ------------------------------------------------------------------ targets
    def get_targets(self, sampling_results, rcnn_train_cfg, concat=True, transform_occ=True,
                    num_occ_per_tracklet=-1):
        """ococc_bbox_head.py:1045-1163 (concat=True form).  The reference computes the targets tracklet by tracklet
        (``_get_target_single``, ~60 tiny launches each) and concatenates; every step of that is row-wise, so here the
        rows of all tracklets are concatenated FIRST and the arithmetic runs once (``_get_targets_batched``): same
        values, a few dozen launches per batch.  Tracklets whose occupancy samples differ in number fall back to the
        per-tracklet form."""
        ks = {tuple(r.occ_labels.shape) for r in sampling_results if len(r.pos_gt_bboxes) > 0}
        if len(ks) <= 1 and all(r.occ_labels is not None for r in sampling_results if len(r.pos_gt_bboxes) > 0):
            return self._get_targets_batched(sampling_results, rcnn_train_cfg, transform_occ, num_occ_per_tracklet)
        return self._get_targets_per_tracklet(sampling_results, rcnn_train_cfg, transform_occ, num_occ_per_tracklet)

    def _get_targets_batched(self, sampling_results, cfg, transform_occ, num_occ_per_tracklet):
        from .tracklet import host_index_many
        dev = sampling_results[0].iou.device
        n_i = [int(r.iou.size(0)) for r in sampling_results]           # samples per tracklet (host numbers)
        p_i = [int(r.pos_gt_bboxes.size(0)) for r in sampling_results]  # of which positives: the first p_i rows
        N, P = sum(n_i), sum(p_i)
        ious = torch.cat([r.iou for r in sampling_results], 0)
        pos_gt_labels = torch.cat([r.pos_gt_labels for r in sampling_results], 0)
        pos_bboxes = torch.cat([r.pos_bboxes for r in sampling_results], 0)
        pos_gt_bboxes = torch.cat([r.pos_gt_bboxes for r in sampling_results], 0)
        if pos_gt_bboxes.size(1) in (9, 10):
            pos_bboxes, pos_gt_bboxes = pos_bboxes[:, :7], pos_gt_bboxes[:, :7]
        starts = [sum(n_i[:t]) for t in range(len(n_i))]
        # every index list of this function is known on the host: two staging buffers (int64 / int32), two copies
        n_occ = [(min(num_occ_per_tracklet, p) if num_occ_per_tracklet > 0 else p) for p in p_i]
        pstart = [sum(p_i[:t]) for t in range(len(p_i))]
        sel_trk = [t for t in range(len(p_i)) for _ in range(n_occ[t])]
        live = [t for t in range(len(p_i)) if p_i[t] > 0]
        slot = {t: k for k, t in enumerate(live)}
        pos_rows_host = [starts[t] + j for t in range(len(n_i)) for j in range(p_i[t])]
        occ_rows_host = [starts[t] + p_i[t] - n_occ[t] + j for t in range(len(p_i)) for j in range(n_occ[t])]
        pos_rows, sel_pos, occ_rows, sel_slot = host_index_many([
            [starts[t] + j for t in range(len(n_i)) for j in range(p_i[t])],
            [pstart[t] + p_i[t] - n_occ[t] + j for t in range(len(p_i)) for j in range(n_occ[t])],
            [starts[t] + p_i[t] - n_occ[t] + j for t in range(len(p_i)) for j in range(n_occ[t])],
            [slot[t] for t in sel_trk]], dev)
        bbox_target_batch_idx, occ_target_batch_idx = host_index_many([
            [t for t in range(len(n_i)) for _ in range(p_i[t])], sel_trk], dev, torch.int32)
        all_gt = torch.full((N,), -1, dtype=pos_gt_labels.dtype, device=dev).index_copy(0, pos_rows, pos_gt_labels)
        label, label_weights = self._soft_label(ious, all_gt, cfg)
        reg_mask = torch.zeros(N, dtype=torch.long, device=dev).index_fill(0, pos_rows, 1)
        bbox_weights = (reg_mask > 0).float()
        cw = cfg.get('class_wise_box_weights', None)
        if cw is not None:
            for i in range(self.num_classes):
                bbox_weights = torch.where(all_gt == i, bbox_weights * cw[i], bbox_weights)
        occ_reg_mask = torch.zeros_like(reg_mask)
        if P > 0:
            bbox_targets = self._canonical_box_targets(pos_bboxes, pos_gt_bboxes)
            occ_reg_mask = occ_reg_mask.index_fill(0, occ_rows, 1)
            occ_all = torch.stack([sampling_results[t].occ_labels for t in live], 0)      # [B', K, 4]
            assert occ_all.dim() == 3 and occ_all.size(2) == 4
            scores_all = torch.stack([sampling_results[t].occ_scores.reshape(-1)[0] for t in live], 0).float()
            with torch.no_grad():
                gt_smp, roi_smp = pos_gt_bboxes[sel_pos], pos_bboxes[sel_pos]
                picked = occ_all[sel_slot]
                roi_local_xyz, gt_occ = picked[..., 0:3], picked[..., 3:4]
                if transform_occ:
                    roi_local_xyz = points_box_to_box(roi_local_xyz, gt_smp, roi_smp)
                occ_score = scores_all[sel_slot]
            pos_gt_bboxes_occ = gt_smp
        else:
            bbox_targets = pos_gt_bboxes.new_empty((0, 7))
            roi_local_xyz = pos_gt_bboxes.new_zeros(0, 0, 3)
            occ_score, gt_occ = pos_gt_bboxes.new_zeros(0), pos_gt_bboxes.new_zeros(0, 1)
            occ_target_batch_idx = torch.zeros(0, dtype=torch.int32, device=dev)
            pos_gt_bboxes_occ = pos_gt_bboxes.new_empty((0, 7))
        label_weights = label_weights / torch.clamp(label_weights.sum(), min=1.0)
        bbox_weights = bbox_weights / torch.clamp(bbox_weights.sum(), min=1.0)
        # (which rows the two masks set, as host lists: with the host copy of the non-empty mask the loss counts its
        # positives without a read-back)
        reg_mask._ococc_rows, occ_reg_mask._ococc_rows = pos_rows_host, (occ_rows_host if P > 0 else [])
        return (label, bbox_targets, bbox_target_batch_idx, pos_gt_bboxes, pos_gt_labels, reg_mask, label_weights,
                bbox_weights, roi_local_xyz, gt_occ, occ_score, occ_reg_mask, occ_target_batch_idx, pos_gt_bboxes_occ)

    def _soft_label(self, ious, all_gt, cfg):
        """get_multi_class_soft_label on rows whose class (-1 = not a positive) is already laid out."""
        pos_thrs, neg_thrs = cfg['cls_pos_thr'], cfg['cls_neg_thr']
        if isinstance(pos_thrs, float):
            pos_thrs, neg_thrs = [pos_thrs] * self.num_classes, [neg_thrs] * self.num_classes
        all_label = ious.new_zeros(ious.size(0))
        for i in range(self.num_classes):
            # 1 above the positive threshold, 0 below the negative one, linear in between: the same values as the three
            # selects of get_multi_class_soft_label (the interpolation is exactly 1 / 0 at the thresholds; clamping leaves
            # the values in between as they are), in three launches
            lab = ((ious - neg_thrs[i]) / (pos_thrs[i] - neg_thrs[i])).clamp(0.0, 1.0)
            all_label = torch.where(all_gt == i, lab, all_label)
        label_weights = (all_label >= 0).float()
        cw = cfg.get('class_wise_cls_weights', None)
        if cw is not None:
            for i in range(self.num_classes):
                label_weights = torch.where(all_gt == i, label_weights * cw[i], label_weights)
        return all_label, label_weights

    def _canonical_box_targets(self, pos_bboxes, pos_gt_bboxes):
        """GT boxes in the canonical frame of their RoIs -> coder deltas (ococc_bbox_head.py:1190-1222)."""
        if type(self.bbox_coder).__name__ == 'DeltaXYZWLHRBBoxCoder' and getattr(self.bbox_coder, 'code_size', 7) == 7 \
                and pos_bboxes.shape[1] == 7 and pos_gt_bboxes.shape == pos_bboxes.shape and _plain(pos_bboxes, pos_gt_bboxes):
            from . import _lib as L   # the frame change and the coder in one launch (csrc/target_ops.hip)
            (rb, ldr), (gb, ldg) = _rows7(pos_bboxes), _rows7(pos_gt_bboxes)
            out = torch.empty((rb.shape[0], 7), dtype=torch.float32, device=rb.device)
            L.check(L.lib.ococc_roi_box_targets_f32(L.ptr(rb), ldr, L.ptr(gb), ldg, rb.shape[0], L.ptr(out), L.stream()),
                    'roi_box_targets')
            return out
        gt_ct = pos_gt_bboxes.clone().detach()
        roi_center = pos_bboxes[..., 0:3]
        roi_ry = pos_bboxes[..., 6] % (2 * np.pi)
        gt_ct[..., 0:3] -= roi_center
        gt_ct[..., 6] -= roi_ry
        gt_ct[..., 0:3] = rotation_3d_in_axis(gt_ct[..., 0:3].unsqueeze(1), -(roi_ry + np.pi / 2), axis=2).squeeze(1)
        ry = gt_ct[..., 6] % (2 * np.pi)
        opposite = (ry > np.pi * 0.5) & (ry < np.pi * 1.5)
        ry = torch.where(opposite, (ry + np.pi) % (2 * np.pi), ry)
        ry = torch.where(ry > np.pi, ry - np.pi * 2, ry)
        gt_ct[..., 6] = torch.clamp(ry, min=-np.pi / 2, max=np.pi / 2)
        anchor = pos_bboxes.clone().detach()
        anchor[:, 0:3] = 0
        anchor[:, 6] = 0
        return self.bbox_coder.encode(anchor, gt_ct)

    def _get_targets_per_tracklet(self, sampling_results, rcnn_train_cfg, transform_occ=True, num_occ_per_tracklet=-1):
        """The reference's loop, kept for batches whose tracklets carry different numbers of occupancy samples (and as
        the statement the batched form is tested against)."""
        per = [self._get_target_single(r.pos_bboxes, r.pos_gt_bboxes, r.iou, r.pos_gt_labels, r.occ_labels,
                                       r.occ_scores, cfg=rcnn_train_cfg, transform_occ=transform_occ,
                                       num_occ_per_tracklet=num_occ_per_tracklet) for r in sampling_results]
        (label, bbox_targets, pos_gt_bboxes, reg_mask, label_weights, bbox_weights, roi_local_xyz, gt_occ,
         occ_score, occ_reg_mask, pos_gt_bboxes_occ) = [list(x) for x in zip(*per)]
        pos_gt_labels = torch.cat([r.pos_gt_labels for r in sampling_results], 0)
        label = torch.cat(label, 0)
        bbox_target_batch_idx = torch.cat([t.new_ones(len(t), dtype=torch.int) * i for i, t in enumerate(bbox_targets)])
        occ_target_batch_idx = torch.cat([t.new_ones(len(t), dtype=torch.int) * i
                                          for i, t in enumerate(pos_gt_bboxes_occ)])
        bbox_targets = torch.cat(bbox_targets, 0)
        pos_gt_bboxes = torch.cat(pos_gt_bboxes, 0)
        pos_gt_bboxes_occ = torch.cat(pos_gt_bboxes_occ, 0)
        reg_mask = torch.cat(reg_mask, 0)
        occ_reg_mask = torch.cat(occ_reg_mask, 0)
        label_weights = torch.cat(label_weights, 0)
        label_weights = label_weights / torch.clamp(label_weights.sum(), min=1.0)
        bbox_weights = torch.cat(bbox_weights, 0)
        bbox_weights = bbox_weights / torch.clamp(bbox_weights.sum(), min=1.0)
        if len(pos_gt_bboxes_occ) > 0:
            pos_roi_local_xyz = torch.cat([e for e in roi_local_xyz if e is not None], 0)
            occ_score = torch.cat([e for e in occ_score if e is not None], 0)
            gt_occ = torch.cat([e for e in gt_occ if e is not None], 0)
        else:
            pos_roi_local_xyz = pos_gt_bboxes.new_zeros(0, 0, 3)
            occ_score = pos_gt_bboxes.new_zeros(0)
            gt_occ = pos_gt_bboxes.new_zeros(0, 1)
        return (label, bbox_targets, bbox_target_batch_idx, pos_gt_bboxes, pos_gt_labels, reg_mask, label_weights,
                bbox_weights, pos_roi_local_xyz, gt_occ, occ_score, occ_reg_mask, occ_target_batch_idx,
                pos_gt_bboxes_occ)

    def _get_target_single(self, pos_bboxes, pos_gt_bboxes, ious, pos_labels, occ_label, occ_score, cfg,
                           transform_occ=True, num_occ_per_tracklet=-1):
        """Canonical-frame box deltas, soft IoU labels and the occupancy query points moved
        from the GT-box frame to the RoI frame (ococc_bbox_head.py:1165-1309)."""
        if pos_gt_bboxes.size(1) in (9, 10):
            pos_bboxes, pos_gt_bboxes = pos_bboxes[:, :7], pos_gt_bboxes[:, :7]
        label, label_weights = self.get_multi_class_soft_label(ious, pos_labels, cfg)
        reg_mask = pos_bboxes.new_zeros(ious.size(0)).long()
        reg_mask[0:pos_gt_bboxes.size(0)] = 1
        bbox_weights = self.get_class_wise_box_weights((reg_mask > 0).float(), pos_labels, cfg)
        occ_reg_mask = torch.zeros_like(reg_mask)
        if pos_gt_bboxes.size(0) > 0 and reg_mask.numel() > 0:  # == reg_mask.bool().any(), known on the host
            bbox_targets = self._canonical_box_targets(pos_bboxes, pos_gt_bboxes)
            assert occ_label.dim() == 2 and occ_label.size(1) == 4
            smp_pos, gt_occ = occ_label[:, 0:3], occ_label[:, 3:4]
            with torch.no_grad():
                num_gt = pos_gt_bboxes.size(0)
                n_occ = min(num_occ_per_tracklet, num_gt) if num_occ_per_tracklet > 0 else num_gt
                roi_local_xyz = smp_pos[None, ...].repeat(n_occ, 1, 1)
                # the last n_occ of the num_gt positives (arange(num_gt)[-n_occ:] upstream; n_occ >= 1 here)
                gt_smp, roi_smp = pos_gt_bboxes[num_gt - n_occ:num_gt], pos_bboxes[num_gt - n_occ:num_gt]
                occ_reg_mask[num_gt - n_occ:num_gt] = 1
                if transform_occ:
                    roi_local_xyz = points_box_to_box(roi_local_xyz, gt_smp, roi_smp)
                gt_occ = gt_occ[None].repeat(len(gt_smp), 1, 1)
                occ_score = occ_score.repeat(len(gt_smp)).float()
        else:
            bbox_targets = pos_gt_bboxes.new_empty((0, 7))
            roi_local_xyz = gt_occ = occ_score = None
            gt_smp = pos_gt_bboxes.new_empty((0, 7))
        return (label, bbox_targets, pos_gt_bboxes, reg_mask, label_weights, bbox_weights, roi_local_xyz, gt_occ,
                occ_score, occ_reg_mask, gt_smp)

    # ------------------------------------------------------------------ losses
    def loss(self, results_dict, rois, labels, bbox_targets, pos_batch_idx, pos_gt_bboxes, pos_gt_labels, reg_mask,
             label_weights, bbox_weights, pos_roi_local_xyz, gt_occ, occ_scores, occ_reg_mask, occ_pos_batch_idx,
             pos_gt_bboxes_occ, transform_occ=False, roi_frame_inds=None):
        """ococc_bbox_head.py:433-606 (corner loss is off in the ococcnet config and not built)."""
        losses = {}
        cls_score, bbox_pred = results_dict['cls_score'], results_dict['bbox_pred']
        nonempty = results_dict['nonempty_roi_mask']
        n_total = cls_score.shape[0]
        assert n_total > 0 and not self.with_corner_loss
        # (ococc_bbox_head.py:464-477 sets the weights of non-empty RoIs to 1, of empty ones to 0, every box weight to 1 and
        # clears the regression mask of empty RoIs, on clones, with masked assignments: the same tensors in one launch each)
        host_hint = (getattr(reg_mask, '_ococc_rows', None), getattr(occ_reg_mask, '_ococc_rows', None),
                     getattr(nonempty, '_ococc_host', None))
        label_weights = nonempty.to(label_weights.dtype)
        bbox_weights = torch.ones_like(bbox_weights)
        reg_mask = torch.where(nonempty, reg_mask, 0)
        cls_avg = n_total * 1.0
        if self.train_cfg.get('sync_cls_avg_factor', False):
            cls_avg = reduce_mean(torch.full((1,), cls_avg, dtype=bbox_weights.dtype, device=bbox_weights.device))
        losses['loss_rcnn_cls'] = self.loss_cls(cls_score.view(-1), labels, label_weights, avg_factor=cls_avg)
        pos_inds = reg_mask > 0
        losses['num_pos_rois'] = pos_inds.sum().float()
        losses['num_neg_rois'] = (reg_mask <= 0).sum().float()
        # the ONE read-back of the loss: how many RoIs are positive for the box loss and for the occupancy loss (the row
        # lists below and in loss_occ are then built at a known size)
        occ_reg_mask = torch.where(nonempty, occ_reg_mask, 0)
        if HOST_POSITIVE_COUNTS and all(h is not None for h in host_hint) and len(host_hint[2]) == n_total:
            # (the rows the masks set and which RoIs have points are both known on the host: no read-back)
            n_pos = sum(1 for r in host_hint[0] if host_hint[2][r])
            n_occ = sum(1 for r in host_hint[1] if host_hint[2][r])
        else:
            n_pos, n_occ = torch.stack([pos_inds.sum(), (occ_reg_mask > 0).sum()]).tolist()
        pos_rows = _true_rows(pos_inds, n_pos)
        reg_avg = pos_rows.numel()
        if self.train_cfg.get('sync_reg_avg_factor', False):
            reg_avg = reduce_mean(torch.full((1,), float(reg_avg), dtype=bbox_weights.dtype, device=bbox_weights.device))
        if pos_rows.numel() == 0:
            losses['loss_rcnn_bbox'] = bbox_pred.sum() * 0
        else:
            pos_pred = bbox_pred[pos_rows]
            bbox_targets = self.filter_pos_assigned_but_empty_rois(bbox_targets, pos_batch_idx, pos_inds,
                                                                   rois[:, 0].int(), count=n_pos)
            w = bbox_weights[pos_rows].view(-1, 1).repeat(1, pos_pred.shape[-1])
            code_weights = self.train_cfg.get('rcnn_code_weights', None)
            if code_weights is not None:
                w = w * const_tensor(code_weights, w.device, w.dtype)[None, :]
            assert pos_pred.size(0) == bbox_targets.size(0)
            losses['loss_rcnn_bbox'] = self.loss_bbox(pos_pred, bbox_targets, w, avg_factor=reg_avg)
        losses.update(self.loss_occ(rois, results_dict['fused_roi_feats'], results_dict['ori_roi_feats'],
                                    occ_pos_batch_idx, pos_gt_bboxes_occ, occ_reg_mask, nonempty,
                                    pos_roi_local_xyz, gt_occ, occ_scores, transform_occ=transform_occ,
                                    roi_frame_inds=roi_frame_inds, num_pos=n_occ))
        return losses

    def loss_occ(self, rois, roi_features, ori_roi_feats, pos_batch_idx, pos_gt_bboxes, reg_mask,
                 nonempty_roi_mask, gt_smp_local_coords, gt_smp_occ_labels, gt_occ_label_scores,
                 transform_occ=False, roi_frame_inds=None, do_aug=False, num_pos=None):
        """ococc_bbox_head.py:608-811 (default train_cfg switches of ococcnet.py: no residual /
        contrastive / outside / observed-feature variants)."""
        losses = {}
        decoder = self.occ_ae_head.occ_decoder
        reg_mask[~nonempty_roi_mask] = 0
        pos_inds = reg_mask > 0
        num_occupied = (gt_smp_occ_labels == 1).sum().float()
        losses['num_occupied'] = num_occupied
        losses['num_free'] = gt_smp_occ_labels.numel() - num_occupied
        pos_rows = _true_rows(pos_inds, num_pos)   # (``num_pos``: the caller has counted them; else one read-back)
        num_pos = pos_rows.numel()
        if pos_rows.numel() == 0:
            idx = torch.arange(roi_features.size(0), device=roi_features.device)
            losses['loss_rcnn_occ'] = decoder(roi_features, roi_features.new_zeros(roi_features.size(0), 3), idx) * 0
            for k in ('recall_neg', 'recall_pos', 'precision_neg', 'precision_pos'):
                losses[k] = roi_features.new_ones(1)
            return losses
        pos_roi_features = roi_features[pos_rows]  # [M, D]
        rb = rois[:, 0].int()
        occ_targets = self.filter_pos_assigned_but_empty_rois(gt_smp_occ_labels, pos_batch_idx, pos_inds, rb, count=num_pos)
        occ_smp_xyz = self.filter_pos_assigned_but_empty_rois(gt_smp_local_coords, pos_batch_idx, pos_inds, rb)
        pos_gt_bboxes = self.filter_pos_assigned_but_empty_rois(pos_gt_bboxes, pos_batch_idx, pos_inds, rb)
        if transform_occ:
            pr = rois[pos_rows][:, 1:]
            with torch.no_grad():
                occ_smp_xyz = points_box_to_box(occ_smp_xyz, pos_gt_bboxes, pr)
        if do_aug:
            occ_smp_xyz[..., 2] = -occ_smp_xyz[..., 2]
        scores = self.filter_pos_assigned_but_empty_rois(gt_occ_label_scores, pos_batch_idx, pos_inds, rb)
        M, K, _ = occ_targets.shape
        occ_weights = (scores > self.occ_label_thresh).to(scores.dtype).view(M, 1).repeat(1, K)
        # decoder with (features, points, RoI index): no [M,K,D] copies (reference :711,:740)
        idx = torch.arange(M, device=rois.device).repeat_interleave(K)
        occ_preds = decoder(pos_roi_features, occ_smp_xyz.reshape(M * K, 3), idx)
        occ_labels = (occ_targets[..., -1] == 1).long()
        losses['loss_rcnn_occ'] = self.loss_occ_comp(occ_preds.view(-1), occ_labels.view(-1), occ_weights.view(-1))
        with torch.no_grad():
            pred_cls = decoder.get_cls_from_pred(occ_preds.view(-1, 1)) if decoder.cls_dim == 1 \
                else decoder.get_cls_from_pred(occ_preds)
            # (the same counts with the validity mask ANDed in instead of eight boolean compactions and their read-backs)
            lab, valid, pred_cls = occ_labels.view(-1), occ_weights.view(-1) > 0, pred_cls.view(-1)
            l0, l1, p0, p1 = (lab == 0) & valid, (lab == 1) & valid, (pred_cls == 0) & valid, (pred_cls == 1) & valid
            # ... and the six counts from one reduction, the four ratios from one division
            c = torch.stack([l0 & p0, l1 & p1, l0, l1, p0, p1]).sum(1)
            # (index tensors from the constant cache: a Python list as an index is a pageable host-to-device copy = a host sync)
            num, den = const_tensor((0, 1, 0, 1), c.device, torch.long), const_tensor((2, 3, 4, 5), c.device, torch.long)
            ratios = c[num] / (c[den] + 1e-6)
            losses['recall_neg'], losses['recall_pos'], losses['precision_neg'], losses['precision_pos'] = ratios.unbind(0)
        return losses
