"""The voxel encoders in front of the SST backbone -- host mirror of mmdet3d/models/voxel_encoders/voxel_encoder.py:
DynamicSimpleVFE (:53-89), DynamicVFE (:92-299) and its DynamicVFELayer (voxel_encoders/utils.py:107-144), the `type`s
the reference's SST configs name (configs/sst/sst_waymoD5_1x_3class_8heads.py:35-45); and the encoders of the HARD
voxel layout ([voxels, max_points, C] + points per voxel, what `voxel.hard_voxelize` emits): HardSimpleVFE (:18-50),
HardVFE (:301-500) with VFELayer and get_paddings_indicator (utils.py:8-104); DynamicScatterVFE and
DynamicRangeScatterVFE (:503-683).  (SIRLayer, the DynamicVFE subclass
OcOccNet uses, lives in sir.py.)

The reference groups the points of a voxel with one DynamicScatter per use and maps voxel rows back to points through
a dense canvas of the whole range (map_voxel_center_to_point, :179-215: batch x Z x Y x X int64 per call).  Here the
grouping is done ONCE per forward -- the unique (b, z, y, x) rows and the inverse map of sst_ops.unique_with_inverse --
and serves the cluster mean, every layer's pooled features and every gather back to the points."""
import torch
from torch import nn

from .registry import VOXEL_ENCODERS, build_norm_layer
from .sst.sst_ops import scatter_v2


def _drop_invalid_voxels(voxel_feats, voxel_coors):
    """The reference's DynamicVFE / DynamicSimpleVFE pool through DynamicScatter, which sets every row with a negative
    coordinate to (-1, ..., -1) and slices that group off its outputs (scatter_points_cuda.cu:202-209): out-of-range
    points of dynamic voxelisation (coors -1) produce no voxel.  scatter_v2 keeps such rows as groups of their own (as
    the reference's scatter_v2 does, for DynamicScatterVFE), so they are removed here.  One read-back, like the
    reference's unique."""
    keep = (voxel_coors >= 0).all(1)
    if bool(keep.all()):
        return voxel_feats, voxel_coors
    return voxel_feats[keep], voxel_coors[keep]


class DynamicVFELayer(nn.Module):
    """Linear(no bias) -> norm -> ReLU (utils.py:107-144)"""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01)):
        super().__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        return torch.relu(self.norm(self.linear(inputs)))


@VOXEL_ENCODERS.register_module()
class DynamicSimpleVFE(nn.Module):
    """mean of the points of every voxel (:53-89)"""

    def __init__(self, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        self.voxel_size, self.point_cloud_range = voxel_size, point_cloud_range
        self.fp16_enabled = False

    @torch.no_grad()
    def forward(self, features, coors):
        return _drop_invalid_voxels(*scatter_v2(features, coors, 'mean', return_inv=False))


@VOXEL_ENCODERS.register_module()
class DynamicVFE(nn.Module):
    """Point features [xyz..., offset to the voxel's point mean, offset to the voxel centre, (range)] through
    ``len(feat_channels)`` DynamicVFELayers, each followed by a max (or mean) over the voxel whose result is handed back
    to the points for the next layer (:92-299).  Constructor arguments, attribute and parameter names as there."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False):
        super().__init__()
        assert len(feat_channels) > 0
        if fusion_layer is not None:
            raise NotImplementedError('image fusion layers are outside this package')
        in_channels += 3 * (int(with_cluster_center) + int(with_voxel_center) + int(with_distance))   # (as there: + 3 each)
        self.in_channels = in_channels
        self._with_distance, self._with_cluster_center = with_distance, with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.vx, self.vy, self.vz = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = point_cloud_range
        self.mode = mode
        chans = [self.in_channels] + list(feat_channels)
        self.vfe_layers = nn.ModuleList(
            [DynamicVFELayer(chans[i] * (2 if i > 0 else 1), chans[i + 1], norm_cfg) for i in range(len(chans) - 1)])
        self.num_vfe = len(self.vfe_layers)
        self.fusion_layer = None

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None):
        """features [N, C] (xyz first), coors [N, 4] (b, z, y, x) -> (voxel_feats [V, feat_channels[-1]], voxel_coors
        [V, 4]) in sorted (b, z, y, x) order, or the per-point features with ``return_point_feats``."""
        features_ls = [features]
        voxel_coors = unq_inv = None
        if self._with_cluster_center:
            voxel_mean, voxel_coors, unq_inv = scatter_v2(features, coors, 'mean')
            features_ls.append(features[:, :3] - voxel_mean.to(features.dtype)[unq_inv.long(), :3])
        if self._with_voxel_center:
            c = coors.to(features.dtype)
            features_ls.append(torch.stack([features[:, 0] - (c[:, 3] * self.vx + self.x_offset),
                                            features[:, 1] - (c[:, 2] * self.vy + self.y_offset),
                                            features[:, 2] - (c[:, 1] * self.vz + self.z_offset)], 1))
        if self._with_distance:
            features_ls.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        features = torch.cat(features_ls, dim=-1)
        reduce = 'max' if self.mode == 'max' else 'mean'
        voxel_feats = point_feats = None
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(features)
            voxel_feats, voxel_coors, unq_inv = scatter_v2(point_feats, coors, reduce, unq_inv=unq_inv, new_coors=voxel_coors)
            if i != len(self.vfe_layers) - 1:
                features = torch.cat([point_feats, voxel_feats.to(point_feats.dtype)[unq_inv.long()]], dim=1)
        if self.return_point_feats:
            return point_feats
        # (the invalid points still took part in the norm layers' batch statistics above, as in the reference)
        return _drop_invalid_voxels(voxel_feats, voxel_coors)


@VOXEL_ENCODERS.register_module()
class DynamicScatterVFE(DynamicVFE):
    """DynamicVFE as the FSD-style configs spell it (:503-612): the cluster offset divided by ``rel_dist_scaler``, the
    pooling ``mode`` handed to scatter_v2 as written ('max' / 'avg' / 'sum'), optionally the inverse map as a third
    result.  ``unique_once`` is accepted and has nothing left to switch: the rows are grouped once per forward here
    in any case (module docstring)."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=1.0, unique_once=False):
        super().__init__(in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center, voxel_size,
                         point_cloud_range, norm_cfg, mode, fusion_layer, return_point_feats)
        self.scatter = self.vfe_scatter = self.cluster_scatter = None
        self.rel_dist_scaler = rel_dist_scaler
        self.unique_once = unique_once

    def map_voxel_center_to_point(self, voxel_mean, voxel2point_inds):
        return voxel_mean[voxel2point_inds.long()]

    def _voxel_origin(self, features, extra):
        """lower corner + half a cell of the grid the voxel indices count from, per axis (x, y, z)"""
        return self.x_offset, self.y_offset, self.z_offset

    def _scatter_forward(self, features, coors, extra, return_inv):
        parts = [features]
        voxel_coors = unq_inv = None
        if self._with_cluster_center:
            mean, voxel_coors, unq_inv = scatter_v2(features[:, :3], coors, 'avg')
            parts.append((features[:, :3] - self.map_voxel_center_to_point(mean.to(features.dtype), unq_inv)) / self.rel_dist_scaler)
        if self._with_voxel_center:
            c = coors.to(features.dtype)
            ox, oy, oz = self._voxel_origin(features, extra)
            parts.append(torch.stack([features[:, 0] - (c[:, 3] * self.vx + ox), features[:, 1] - (c[:, 2] * self.vy + oy),
                                      features[:, 2] - (c[:, 1] * self.vz + oz)], 1))
        if self._with_distance:
            parts.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        x = torch.cat(parts, dim=-1)
        voxel_feats = point_feats = None
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(x)
            voxel_feats, voxel_coors, unq_inv = scatter_v2(point_feats, coors, self.mode, unq_inv=unq_inv, new_coors=voxel_coors)
            if i != len(self.vfe_layers) - 1:
                x = torch.cat([point_feats, self.map_voxel_center_to_point(voxel_feats.to(point_feats.dtype), unq_inv)], dim=1)
        if self.return_point_feats:
            return point_feats
        return (voxel_feats, voxel_coors, unq_inv) if return_inv else (voxel_feats, voxel_coors)

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, return_inv=False):
        return self._scatter_forward(features, coors, None, return_inv)


@VOXEL_ENCODERS.register_module()
class DynamicRangeScatterVFE(DynamicScatterVFE):
    """the same with the grid's origin given per point (``pts_min_bounds`` [N, 3] (x, y, z): a grid per object), taken
    as written there -- WITHOUT the half-cell the fixed-range classes add (:615-683)"""

    def _voxel_origin(self, features, extra):
        return extra[:, 0], extra[:, 1], extra[:, 2]

    def forward(self, features, coors, pts_min_bounds, points=None, img_feats=None, img_metas=None, return_inv=False):
        return self._scatter_forward(features, coors, pts_min_bounds, return_inv)


def get_paddings_indicator(actual_num, max_num, axis=0):
    """[len(actual_num), max_num] bool: slot j of row i holds a point iff j < actual_num[i] (utils.py:8-28; ``axis`` is the
    dimension of ``actual_num`` the slots are put behind)."""
    slots = torch.arange(max_num, dtype=torch.int, device=actual_num.device)
    shape = [1] * (actual_num.dim() + 1)
    shape[axis + 1] = -1
    return actual_num.unsqueeze(axis + 1).int() > slots.view(shape)


class VFELayer(nn.Module):
    """Linear(no bias) -> norm over the channels -> ReLU on [voxels, slots, C]; then, by flags: the point features
    (``max_out=False``), the maximum over the slots [voxels, C'] (``cat_max=False``) or the point features with that
    maximum appended to every slot [voxels, slots, 2 C'] (utils.py:31-104)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), max_out=True,
                 cat_max=True):
        super().__init__()
        self.fp16_enabled = False
        self.cat_max, self.max_out = cat_max, max_out
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        v, m, _ = inputs.shape
        x = self.linear(inputs)
        # (the norm sees channels in dimension 1 of a 3-D tensor there; over [v * m, C'] rows the statistics are the same)
        pointwise = torch.relu(self.norm(x.reshape(v * m, -1)).reshape(v, m, -1))
        if not self.max_out:
            return pointwise
        top = pointwise.amax(dim=1, keepdim=True)
        if not self.cat_max:
            return top.squeeze(1)
        return torch.cat([pointwise, top.expand(-1, m, -1)], dim=2)


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    """mean of the first ``num_features`` columns over the points a voxel holds (:18-50)"""

    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors):
        total = features[:, :, :self.num_features].sum(dim=1)
        return (total / num_points.to(features.dtype).view(-1, 1)).contiguous()


@VOXEL_ENCODERS.register_module()
class HardVFE(nn.Module):
    """DynamicVFE's decorations and layer stack on the hard layout: features [voxels, max_points, C] with zero rows
    behind ``num_points`` points, coors [voxels, 4] (b, z, y, x) -> [voxels, feat_channels[-1]] (:301-500).  The last
    layer returns the maximum alone; the others hand [point, maximum] on."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False):
        super().__init__()
        assert len(feat_channels) > 0
        if fusion_layer is not None:
            raise NotImplementedError('image fusion layers are outside this package')
        in_channels += 3 * (int(with_cluster_center) + int(with_voxel_center) + int(with_distance))   # (as there: + 3 each)
        self.in_channels = in_channels
        self._with_distance, self._with_cluster_center = with_distance, with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.vx, self.vy, self.vz = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = point_cloud_range
        chans = [self.in_channels] + list(feat_channels)
        last = len(chans) - 2
        self.vfe_layers = nn.ModuleList(
            [VFELayer(chans[i] * (2 if i > 0 else 1), chans[i + 1], norm_cfg=norm_cfg, max_out=True, cat_max=i != last)
             for i in range(len(chans) - 1)])
        self.num_vfe = len(self.vfe_layers)
        self.fusion_layer = None

    def forward(self, features, num_points, coors, img_feats=None, img_metas=None):
        parts = [features]
        xyz = features[:, :, :3]
        if self._with_cluster_center:
            parts.append(xyz - xyz.sum(dim=1, keepdim=True) / num_points.to(features.dtype).view(-1, 1, 1))
        if self._with_voxel_center:
            c = coors.to(features.dtype)
            centre = torch.stack([c[:, 3] * self.vx + self.x_offset, c[:, 2] * self.vy + self.y_offset,
                                  c[:, 1] * self.vz + self.z_offset], 1)
            parts.append(xyz - centre.unsqueeze(1))
        if self._with_distance:
            parts.append(torch.norm(xyz, 2, 2, keepdim=True))
        x = torch.cat(parts, dim=-1)
        # the decorations of the empty slots are not zero: cleared here, as there
        x = x * get_paddings_indicator(num_points, x.shape[1], axis=0).unsqueeze(-1).to(x.dtype)
        for vfe in self.vfe_layers:
            x = vfe(x)
        return x
