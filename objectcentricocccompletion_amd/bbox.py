"""Box helpers on the hot path -- host mirror of rotation_3d_in_axis
(mmdet3d/core/bbox/structures/utils.py:21-61) and DeltaXYZWLHRBBoxCoder
(mmdet3d/core/bbox/coders/delta_xyzwhlr_bbox_coder.py:8-90).  Tiny elementwise math."""
import torch

from .registry import BBOX_CODERS


def limit_period(val, offset=0.5, period=3.141592653589793):
    return val - torch.floor(val / period + offset) * period


def _rows7(t):
    """a [n, >= 7] f32 device tensor with unit column stride as (tensor, row stride); a copy only if it is not that"""
    if not (t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
        t = t.float().contiguous()
    return t, t.stride(0)


def _plain(*tensors):
    """device f32 tensors nobody differentiates through: the single-launch forms apply"""
    return all(t.is_cuda and t.dtype == torch.float32 and not t.requires_grad for t in tensors)


def rotation_3d_in_axis(points, angles, axis=0):
    """points [N,M,3] rotated by angles [N] about `axis`; note the reference multiplies by the
    TRANSPOSED matrix (einsum 'aij,jka->aik'), i.e. a clockwise turn for axis 2."""
    if (axis == 2 or axis == -1) and points.dim() == 3 and points.shape[-1] == 3 and angles.dim() == 1 \
            and angles.shape[0] == points.shape[0] and _plain(points, angles):
        from . import _lib as L   # one launch instead of a dozen (csrc/target_ops.hip)
        p, a = points.contiguous(), angles.contiguous()
        out = torch.empty_like(p)
        L.check(L.lib.ococc_rotate_z_f32(L.ptr(p), L.ptr(a), p.shape[0], p.shape[1], L.ptr(out), L.stream()), 'rotate_z')
        return out
    rot_sin, rot_cos = torch.sin(angles), torch.cos(angles)
    ones, zeros = torch.ones_like(rot_cos), torch.zeros_like(rot_cos)
    if axis == 1:
        rows = [[rot_cos, zeros, -rot_sin], [zeros, ones, zeros], [rot_sin, zeros, rot_cos]]
    elif axis == 2 or axis == -1:
        rows = [[rot_cos, -rot_sin, zeros], [rot_sin, rot_cos, zeros], [zeros, zeros, ones]]
    elif axis == 0:
        rows = [[zeros, rot_cos, -rot_sin], [zeros, rot_sin, rot_cos], [ones, zeros, zeros]]
    else:
        raise ValueError(f'axis should in range [0, 1, 2], got {axis}')
    rot_mat_T = torch.stack([torch.stack(r) for r in rows])
    return torch.einsum('aij,jka->aik', (points, rot_mat_T))


def points_box_to_box(xyz, from_boxes, to_boxes):
    """xyz [N, M, 3] given in the (gravity-centred) frame of from_boxes[i] -> the frame of to_boxes[i]: the chain of
    ococc_bbox_head.py:1279-1290 / 714-724 (GT-box frame -> ego frame -> RoI frame); boxes [N, >= 7]."""
    if xyz.dim() == 3 and xyz.shape[-1] == 3 and from_boxes.shape[0] == xyz.shape[0] == to_boxes.shape[0] \
            and _plain(xyz, from_boxes, to_boxes) and from_boxes.shape[1] >= 7 and to_boxes.shape[1] >= 7:
        from . import _lib as L
        p = xyz.contiguous()
        (fb, ldf), (tb, ldt) = _rows7(from_boxes), _rows7(to_boxes)
        out = torch.empty_like(p)
        L.check(L.lib.ococc_points_box_to_box_f32(L.ptr(p), L.ptr(fb), ldf, L.ptr(tb), ldt, p.shape[0], p.shape[1], L.ptr(out),
                                                  L.stream()), 'points_box_to_box')
        return out
    xyz = rotation_3d_in_axis(xyz, from_boxes[:, 6], axis=2)
    xyz += from_boxes[..., None, 0:3]
    xyz[..., 2] += from_boxes[:, None, 5] / 2   # voxel centres are gravity centred
    xyz -= to_boxes[..., None, :3]
    xyz[..., 2] -= to_boxes[:, None, 5] / 2
    return rotation_3d_in_axis(xyz, -(to_boxes[:, 6]), axis=2)


def box_corners(boxes):
    """[N, 7+] (x, y, z_bottom, dx, dy, dz, yaw) -> [N, 8, 3], the corner order of LiDARInstance3DBoxes.corners
    (mmdet3d/core/bbox/structures/lidar_box3d.py:54-92): (x0y0z0, x0y0z1, x0y1z1, x0y1z0, x1y0z0, x1y0z1, x1y1z1, x1y1z0)
    about the bottom centre, turned about z with rotation_3d_in_axis."""
    unit = boxes.new_tensor([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]])
    unit = unit - boxes.new_tensor([0.5, 0.5, 0.0])
    corners = boxes[:, 3:6].reshape(-1, 1, 3) * unit[None]
    return rotation_3d_in_axis(corners, boxes[:, 6], axis=2) + boxes[:, :3].reshape(-1, 1, 3)


@BBOX_CODERS.register_module()
class DeltaXYZWLHRBBoxCoder(object):
    """(x, y, z_bottom, w, l, h, r) deltas normalised by the anchor diagonal / height."""

    def __init__(self, code_size=7):
        self.code_size = code_size

    @staticmethod
    def encode(src_boxes, dst_boxes):
        xa, ya, za, wa, la, ha, ra, *cas = torch.split(src_boxes, 1, dim=-1)
        xg, yg, zg, wg, lg, hg, rg, *cgs = torch.split(dst_boxes, 1, dim=-1)
        cts = [g - a for g, a in zip(cgs, cas)]
        za = za + ha / 2
        zg = zg + hg / 2
        diagonal = torch.sqrt(la ** 2 + wa ** 2)
        return torch.cat([(xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / ha, torch.log(wg / wa),
                          torch.log(lg / la), torch.log(hg / ha), rg - ra, *cts], dim=-1)

    @staticmethod
    def decode(anchors, deltas):
        xa, ya, za, wa, la, ha, ra, *cas = torch.split(anchors, 1, dim=-1)
        xt, yt, zt, wt, lt, ht, rt, *cts = torch.split(deltas, 1, dim=-1)
        za = za + ha / 2
        diagonal = torch.sqrt(la ** 2 + wa ** 2)
        xg, yg, zg = xt * diagonal + xa, yt * diagonal + ya, zt * ha + za
        lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
        zg = zg - hg / 2
        cgs = [t + a for t, a in zip(cts, cas)]
        return torch.cat([xg, yg, zg, wg, lg, hg, rt + ra, *cgs], dim=-1)


def build_bbox_coder(cfg):
    return BBOX_CODERS.build(cfg)
