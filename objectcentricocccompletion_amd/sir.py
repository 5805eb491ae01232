"""SIR point encoders -- host mirror of SIRLayer (mmdet3d/models/voxel_encoders/
voxel_encoder.py:686-832), DynamicVFELayerV2 (voxel_encoders/utils.py:147-189) and SIR
(mmdet3d/models/backbones/sir.py:16-88).  Same constructor arguments, parameter names
(rel_mlp.<i>.0.weight, vfe_layers.<j>.{linear,norm}.*, block_list.<i>...) and outputs.

Device work per layer: Linear -> cuBLAS-class GEMM (hipBLASLt through torch),
LN(+GELU) -> ococc_layernorm_act_*, scatter max/mean -> ococc_segment_reduce_f32.
"""
import torch
from torch import nn

from ._lib import const_tensor
from .registry import BACKBONES, VOXEL_ENCODERS, build_norm_layer
from .sst.sst_ops import (build_mlp, fuse_norm_act, get_activation_layer, scatter_v2,
                          unique_with_inverse)
from .linear import Linear
from .voxel.scatter_points import gather_rows


class DynamicVFELayerV2(nn.Module):
    """Linear(no bias) -> norm -> act (utils.py:147-189)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 act='relu', dropout=0.0):
        super().__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = Linear(in_channels, out_channels, bias=False)
        self.act = get_activation_layer(act, out_channels)
        self.norm, self.act = fuse_norm_act(self.norm, self.act)
        self.dropout = nn.Dropout(p=dropout) if dropout > 0 else None

    def forward(self, inputs):
        if self.dropout is not None:
            inputs = self.dropout(inputs)
        return self.act(self.norm(self.linear(inputs)))


@VOXEL_ENCODERS.register_module()
class SIRLayer(nn.Module):
    """voxel_encoder.py:686-832.  The reference derives from DynamicVFE only to inherit flags;
    the members it actually uses are kept, with the same names."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_rel_mlp=True, rel_mlp_hidden_dims=[16, ], rel_mlp_in_channel=3,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4),
                 point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=1.0, with_shortcut=True,
                 xyz_normalizer=[1.0, 1.0, 1.0], act='relu', dropout=0.0):
        super().__init__()
        assert len(feat_channels) > 0
        raw_in_channels = in_channels
        # DynamicVFE.__init__ (voxel_encoder.py:136-143) widens in_channels for the decorations
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 3
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.rel_dist_scaler = rel_dist_scaler
        self.mode = mode
        self.with_shortcut = with_shortcut
        self._with_rel_mlp = with_rel_mlp
        self.xyz_normalizer = xyz_normalizer
        if with_rel_mlp:
            # the reference appends to the caller's list (voxel_encoder.py:733); the config
            # loader hands every block its own copy, and so do we
            rel_mlp_hidden_dims = list(rel_mlp_hidden_dims) + [raw_in_channels]  # 'not self.in_channels'
            self.rel_mlp = build_mlp(rel_mlp_in_channel, rel_mlp_hidden_dims, norm_cfg, act=act)
        feat_channels = [self.in_channels] + list(feat_channels)
        vfe_layers = []
        for i in range(len(feat_channels) - 1):
            in_filters = feat_channels[i]
            if i > 0:
                in_filters *= 2
            vfe_layers.append(DynamicVFELayerV2(in_filters, feat_channels[i + 1], norm_cfg, act=act,
                                                dropout=dropout))
        self.vfe_layers = nn.ModuleList(vfe_layers)
        self.num_vfe = len(vfe_layers)

    def forward(self, features, coors, f_cluster=None, points=None, img_feats=None, img_metas=None,
                return_inv=False, return_both=False, unq_inv_once=None, new_coors_once=None):
        xyz_normalizer = const_tensor(self.xyz_normalizer, features.device, features.dtype)
        features_ls = [torch.cat([features[:, :3] / xyz_normalizer[None, :], features[:, 3:]], dim=1)]
        if self.with_shortcut:
            shortcut = features[:, 3:]
        if f_cluster is None:
            voxel_mean, mean_coors, unq_inv = scatter_v2(features[:, :3], coors, mode='avg',
                                                         unq_inv=unq_inv_once, new_coors=new_coors_once)
            points_mean = gather_rows(voxel_mean, unq_inv)
            f_cluster = (features[:, :3] - points_mean[:, :3]) / self.rel_dist_scaler
        else:
            f_cluster = f_cluster / self.rel_dist_scaler
        if self._with_cluster_center:
            features_ls.append(f_cluster / 10.0)
        if self._with_rel_mlp:
            features_ls[0] = features_ls[0] * self.rel_mlp(f_cluster)
        if self._with_distance:
            features_ls.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        features = torch.cat(features_ls, dim=-1)

        voxel_feats_list = []
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(features)
            voxel_feats, voxel_coors, unq_inv = scatter_v2(point_feats, coors, mode=self.mode,
                                                           unq_inv=unq_inv_once, new_coors=new_coors_once)
            voxel_feats_list.append(voxel_feats)
            if i != len(self.vfe_layers) - 1:
                features = torch.cat([point_feats, gather_rows(voxel_feats, unq_inv)], dim=1)
        voxel_feats = torch.cat(voxel_feats_list, dim=1)

        if return_both:
            if self.with_shortcut and point_feats.shape == shortcut.shape:
                point_feats = point_feats + shortcut
            return point_feats, voxel_feats, voxel_coors
        if self.return_point_feats:
            if self.with_shortcut and point_feats.shape == shortcut.shape:
                point_feats = point_feats + shortcut
            return point_feats, voxel_feats
        if return_inv:
            return voxel_feats, voxel_coors, unq_inv
        return voxel_feats, voxel_coors


@BACKBONES.register_module()
class SIR(nn.Module):
    """Stack of SIRLayers (backbones/sir.py:16-88)."""

    def __init__(self, num_blocks=5, in_channels=[], feat_channels=[], rel_mlp_hidden_dims=[],
                 with_rel_mlp=True, with_distance=False, with_cluster_center=False,
                 norm_cfg=dict(type='LN', eps=1e-3), mode='max', xyz_normalizer=[1.0, 1.0, 1.0],
                 act='relu', dropout=0, unique_once=False):
        super().__init__()
        self.num_blocks = num_blocks
        self.unique_once = unique_once
        block_list = []
        for i in range(num_blocks):
            block_list.append(SIRLayer(
                in_channels=in_channels[i], feat_channels=feat_channels[i], with_distance=with_distance,
                with_cluster_center=with_cluster_center, with_rel_mlp=with_rel_mlp,
                rel_mlp_hidden_dims=rel_mlp_hidden_dims[i], with_voxel_center=False,
                voxel_size=[0.1, 0.1, 0.1], point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4],
                norm_cfg=norm_cfg, mode=mode, fusion_layer=None,
                return_point_feats=i != num_blocks - 1, return_inv=False, rel_dist_scaler=10.0,
                xyz_normalizer=xyz_normalizer, act=act, dropout=dropout))
        self.block_list = nn.ModuleList(block_list)

    def forward(self, points, features, coors, f_cluster=None, dims=None):
        if self.unique_once:
            new_coors, unq_inv = unique_with_inverse(coors, dims)
        else:
            new_coors = unq_inv = None
        out_feats = features
        cluster_feat_list = []
        for i, block in enumerate(self.block_list):
            in_feats = torch.cat([points, out_feats], 1)
            if i < self.num_blocks - 1:
                out_feats, out_cluster_feats = block(in_feats, coors, f_cluster, unq_inv_once=unq_inv,
                                                     new_coors_once=new_coors)
            else:
                out_feats, out_cluster_feats, out_coors = block(in_feats, coors, f_cluster, return_both=True,
                                                                unq_inv_once=unq_inv,
                                                                new_coors_once=new_coors)
            cluster_feat_list.append(out_cluster_feats)
        return out_feats, torch.cat(cluster_feat_list, dim=1), out_coors
