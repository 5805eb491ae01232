"""Points-in-box RoI pooling -- host mirror of mmdet3d/ops/dynamic_point_pool_op.py:63-113
(dynamic_point_pool_mixed) and TrackletPointRoIExtractor
(mmdet3d/models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:151-243).
Kernel: ococc_dynamic_point_pool_mixed (rows come back sorted by (RoI, point))."""
import torch
from torch import nn

from . import _lib as L
from .registry import ROI_EXTRACTORS


def dynamic_point_pool_mixed(rois, rois_batch, pts, pts_batch, extra_wlh, max_inbox_point,
                             max_all_pts=200000, return_counts=False, also_read=None):
    """Same arguments / returns as DynamicPointPoolMixedFunction.forward: (out_pts_idx [M] i64,
    out_roi_idx [M] i64, out_pts_feats [M,13] f32).  When nothing is inside any box the
    reference returns one fake row of -1 / zeros (dynamic_point_pool_op.py:91-95); so do we."""
    L.require_device(rois, pts)
    assert len(rois) > 0
    rois = rois.contiguous().float()
    pts = pts.contiguous().float()
    rois_batch = rois_batch.contiguous().to(torch.int32)
    pts_batch = pts_batch.contiguous().to(torch.int32)
    R, N = rois.size(0), pts.size(0)
    dev = pts.device
    out_pts_idx = torch.empty((max_all_pts,), dtype=torch.long, device=dev)
    out_roi_idx = torch.empty((max_all_pts,), dtype=torch.long, device=dev)
    out_feats = torch.empty((max_all_pts, 13), dtype=torch.float32, device=dev)
    meta = torch.zeros((R + 1,), dtype=torch.int32, device=dev)  # [num_out, roi_counts...]
    ws = L.workspace(L.lib.ococc_point_pool_workspace_bytes(N, R), dev)
    L.check(L.lib.ococc_dynamic_point_pool_mixed(
        L.ptr(rois), L.ptr(rois_batch), R, L.ptr(pts), L.ptr(pts_batch), N, L.f3(extra_wlh),
        int(max_inbox_point), int(max_all_pts), L.ptr(out_pts_idx), L.ptr(out_roi_idx),
        L.ptr(out_feats), meta.data_ptr() + 4, meta.data_ptr(), L.ptr(ws), ws.numel(), L.stream()),
        'dynamic_point_pool_mixed')
    # the one read-back (the reference's boolean-mask compaction syncs too); ``also_read`` (a small int32 device tensor of the
    # caller's) rides on it and comes back as a list in also_read.host
    # ... and so does the number of RoIs that received a point: the grouping of the pooled points by RoI
    # (sst_ops.unique_with_inverse) then knows its output size without a read-back of its own
    # (the per-RoI counts come along whole -- R small integers: the loss then knows on the host which of its positive RoIs are empty)
    if also_read is None:
        got = meta.tolist()
    else:
        got = torch.cat([meta, also_read.to(torch.int32)]).tolist()
        also_read.host = got[R + 1:]
    # (the host has just waited for the device: the cheapest place to learn that a grid barrier of a one-launch SIR layer
    # of the step before gave up -- csrc/sir_fused.hip; raises)
    L.check(L.lib.ococc_sir_layer_fused_check(), 'sir_layer barrier check')
    m, roi_counts = int(got[0]), got[1:R + 1]
    nonempty = sum(1 for c in roi_counts if c > 0)
    if m == 0:
        out = (out_pts_idx.new_full((1,), -1), out_roi_idx.new_full((1,), -1), out_feats.new_zeros((1, 13)))
        nonempty = 1   # (the one fake row is a group of its own)
    else:
        out = (out_pts_idx[:m], out_roi_idx[:m], out_feats[:m])
    out[1]._ococc_num_groups = int(nonempty)
    out[1]._ococc_roi_counts = roi_counts
    if return_counts:
        return out + (meta[1:],)
    return out


_KEY_STRIDE = 1 << 16   # frames per batch entry in the (batch, frame) match keys


@ROI_EXTRACTORS.register_module()
class TrackletPointRoIExtractor(nn.Module):
    """Point-wise RoI extractor over tracklets: a point belongs to the RoI of its own
    (batch, frame) (dynamic_point_roi_extractor.py:177-243)."""

    def __init__(self, init_cfg=None, debug=True, extra_wlh=[0, 0, 0], max_inbox_point=512,
                 max_all_point=200000, combined=False):
        super().__init__()
        self.debug = debug
        self.extra_wlh = extra_wlh
        self.max_inbox_point = max_inbox_point
        self.max_all_point = max_all_point
        self.combined = combined

    def forward(self, pts_xyz, batch_inds, pts_frame_inds, rois, roi_frame_inds, max_inbox_point=None):
        assert len(pts_xyz) > 0 and len(batch_inds) > 0 and len(rois) > 0
        seen = None
        if self.combined:
            pts_inds, roi_inds = batch_inds.int(), rois[:, 0].int()
        else:
            # The (batch, frame) keys are only compared for equality: a fixed stride instead of the largest frame index + 1
            # (dynamic_point_roi_extractor.py:186-190 reads both maxima back for it) -- the reference's check that no point
            # lies in a later frame than any RoI is made from the maxima riding on the pooling's own read-back below.
            max_frames = _KEY_STRIDE
            seen = torch.stack([roi_frame_inds.max(), pts_frame_inds.max(), batch_inds.max()])
            pts_inds = (batch_inds * max_frames + pts_frame_inds).int()
            roi_inds = (rois[:, 0].int() * max_frames + roi_frame_inds).int()
        if isinstance(self.max_all_point, (tuple, list)):
            max_all_point = self.max_all_point[0] if self.training else self.max_all_point[1]
        else:
            max_all_point = self.max_all_point
        all_inds, all_roi_inds, info = dynamic_point_pool_mixed(
            rois[..., 1:], roi_inds, pts_xyz, pts_inds, self.extra_wlh, self.max_inbox_point,
            max_all_point, also_read=seen)
        if seen is not None:
            roi_frames, pts_frames, batches = (int(v) + 1 for v in seen.host)
            assert pts_frames <= roi_frames, f'{pts_frames} > {roi_frames}'
            assert roi_frames <= _KEY_STRIDE and batches * _KEY_STRIDE < 2 ** 31, 'frame / batch indices beyond the key range'
        ext_pts_info = dict(local_xyz=info[:, 3:6], boundary_offset=info[:, 6:-1], is_in_margin=info[:, -1])
        if self.debug:
            r = rois[..., 1:][all_roi_inds]
            off = ext_pts_info['boundary_offset']
            assert torch.isclose(pts_xyz[all_inds], info[:, :3]).all()
            assert torch.isclose(off[:, 0] + off[:, 3], r[:, 4]).all()
            assert torch.isclose(off[:, 1] + off[:, 4], r[:, 3]).all()
            assert torch.isclose(off[:, 2] + off[:, 5], r[:, 5]).all()
        return all_inds, all_roi_inds, ext_pts_info
