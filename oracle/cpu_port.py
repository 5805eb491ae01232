"""Run the product's OcOccNet module graph on the CPU by swapping its six HIP-backed leaf operators for torch /
oracle restatements -- the ``cpu_baseline`` of ``bench.py --workload ococcnet`` (kind "port"), and a CPU-side parity
check of the host logic against the reference goldens.  TEST INFRASTRUCTURE ONLY: the product itself has no CPU
path (its operators reject CPU tensors); nothing under objectcentricocccompletion_amd/ imports this file.

Leaf operators and what they restate:
  layer_norm_act            nn.LayerNorm (+ nn.GELU(), exact erf)      sst_ops.py:333-360, occ_base.py:97, layers.py:57-58
  grid_unique               torch.unique(dim=0, sorted), -1 rows dropped   sst_ops.py:155-158 / scatter_points_cuda.cu:199-210
  segment_reduce            torch_scatter scatter_max / scatter(mean|sum)   sst_ops.py:171-174
  gather_rows               feats[inv]                                      voxel_encoder.py:758-760
  dynamic_point_pool_mixed  TorchEx contract, served by the C oracle        dynamic_point_pool_op.py:63-113
  aligned_iou_3d            LiDARInstance3DBoxes.aligned_iou_3d, C oracle    lidar_box3d.py:404-448
"""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


def _layer_norm_act(x, weight, bias, eps=1e-5, act='none', dropout=0.0):
    y = F.layer_norm(x.float(), (x.shape[-1],), weight.float(), bias.float(), eps).to(x.dtype)
    y = F.gelu(y) if act == 'gelu' else y
    return F.dropout(y, dropout, True) if dropout else y


def _grid_unique(coors, dims=None, static=False):
    assert not static
    c = coors if coors.dim() == 2 else coors[:, None]
    keep = (c >= 0).all(1)
    out, inv_k = torch.unique(c[keep], dim=0, return_inverse=True)
    inv = torch.full((c.shape[0],), -1, dtype=torch.int32)
    inv[keep] = inv_k.to(torch.int32)
    counts = torch.bincount(inv_k, minlength=out.shape[0]).to(torch.int32)
    out = out.to(torch.int32)
    return (out[:, 0] if coors.dim() == 1 else out), inv, counts


def _segment_reduce(feats, inv, num_segments, mode, counts=None):
    idx = inv.long()
    keep = idx >= 0
    src, idx = feats[keep], idx[keep]
    ix = idx[:, None].expand_as(src)
    if mode == 'max':
        out = feats.new_full((num_segments, feats.shape[1]), float('-inf')).scatter_reduce(0, ix, src, 'amax', include_self=True)
        return torch.where(torch.isinf(out), torch.zeros_like(out), out)
    red = {'mean': 'mean', 'avg': 'mean', 'sum': 'sum'}[mode]
    return feats.new_zeros((num_segments, feats.shape[1])).scatter_reduce(0, ix, src, red, include_self=False)


def _gather_rows(rows, inv):
    return rows[inv.long()]


def _dynamic_point_pool_mixed(rois, rois_batch, pts, pts_batch, extra_wlh, max_inbox_point, max_all_pts=200000,
                              return_counts=False, also_read=None):
    pi, ri, fe, cnt = O.point_pool(rois.detach().numpy(), rois_batch.numpy(), pts.detach().numpy(), pts_batch.numpy(),
                                   list(extra_wlh), int(max_inbox_point), int(max_all_pts))
    if also_read is not None:   # (the product reads this small tensor back together with its output size)
        also_read.host = [int(v) for v in also_read.tolist()]
    if len(pi) == 0:
        out = (torch.full((1,), -1, dtype=torch.long), torch.full((1,), -1, dtype=torch.long), torch.zeros((1, 13)))
    else:
        out = (torch.from_numpy(pi), torch.from_numpy(ri), torch.from_numpy(fe))
    return out + (torch.from_numpy(cnt),) if return_counts else out


def _aligned_iou_3d(b1, b2):
    return torch.from_numpy(O.aligned_iou3d(b1[:, :7].detach().numpy(), b2[:, :7].detach().numpy()))


@contextlib.contextmanager
def cpu_ops():
    """Inside the block the product's module graph runs on CPU tensors."""
    from objectcentricocccompletion_amd import _lib, norm, point_pool, sir, tracklet
    from objectcentricocccompletion_amd.occ import layers, occ_base
    from objectcentricocccompletion_amd.sst import sst_ops
    patches = [(norm, 'layer_norm_act', _layer_norm_act), (layers, 'layer_norm_act', _layer_norm_act),
               (occ_base, 'layer_norm_act', _layer_norm_act), (sst_ops, 'grid_unique', _grid_unique),
               (sst_ops, 'segment_reduce', _segment_reduce), (sir, 'segment_reduce', _segment_reduce),
               (sir, 'gather_rows', _gather_rows),
               (occ_base, 'gather_rows', _gather_rows),
               (point_pool, 'dynamic_point_pool_mixed', _dynamic_point_pool_mixed),
               (tracklet, 'aligned_iou_3d', _aligned_iou_3d), (_lib, 'require_device', lambda *a, **k: None),
               # the fused per-point / decoder kernels have no CPU stand-in: their op-by-op forms (patched above) run
               (sir, 'POINT_LAYER_KERNEL', False), (occ_base, 'FUSED_MLP', False)]
    saved = [(m, n, getattr(m, n)) for m, n, _ in patches]
    try:
        for m, n, f in patches:
            setattr(m, n, f)
        yield
    finally:
        for m, n, f in saved:
            setattr(m, n, f)


def build_detector_cpu(seed_weights=True):
    """The product's TrackletDetectorOCC (ococcnet config, 66 553 173 parameters) on the CPU, name-hashed weights."""
    from objectcentricocccompletion_amd import heads, point_pool, roi_head  # noqa: F401 (register)
    from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg
    from objectcentricocccompletion_amd.registry import DETECTORS
    from . import synth
    m = DETECTORS.build(ococcnet_model_cfg())
    if seed_weights:
        bh = m.roi_head.bbox_head
        bh.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in bh.state_dict().items()}, seed=0))
    return m
