"""Host mirror of mmdet3d/ops/sst/sst_ops.py: scatter_v2 (:150-181), build_mlp (:333-360),
get_activation / get_activation_layer (:362-391).  The torch.unique + torch_scatter pair of
the reference becomes ococc_grid_unique_i32 + ococc_segment_reduce_f32 (HIP)."""
import torch
import torch.nn as nn

from ..norm import LayerNorm
from ..registry import build_norm_layer
from ..voxel.scatter_points import grid_unique, segment_reduce


def unique_with_inverse(coors, dims=None, return_counts=False):
    """torch.unique(coors, return_inverse=True, dim=0) for integer keys >= -1 (the reference
    feeds RoI / voxel indices where -1 marks "no group" and is kept as a group of its own,
    sorted first -- ococc_bbox_head.py:255-260).  Returns int32 tensors."""
    shifted = coors + 1  # -1 -> 0: the bitmap ranks only non-negative keys
    if dims is not None:
        dims = [int(d) + 1 for d in dims]
    new_coors, inv, counts = grid_unique(shifted, dims)
    new_coors = new_coors - 1
    if return_counts:
        return new_coors, inv, counts
    return new_coors, inv


def scatter_v2(feat, coors, mode, return_inv=True, min_points=0, unq_inv=None, new_coors=None,
               dims=None):
    """sst_ops.py:150-181.  ``dims`` (exclusive upper bound of each key column) is an
    extension that spares the device read-back torch.unique implies."""
    assert feat.size(0) == coors.size(0)
    if mode == 'avg':
        mode = 'mean'
    counts = None
    if unq_inv is None:
        new_coors, unq_inv, counts = unique_with_inverse(coors, dims, return_counts=True)
    else:
        assert new_coors is not None, 'please pass new_coors for interface consistency'
    if min_points > 0:
        if counts is None:
            counts = torch.bincount(unq_inv.long(), minlength=new_coors.size(0))
        valid_mask = counts[unq_inv.long()] >= min_points
        feat = feat[valid_mask]
        coors = coors[valid_mask]
        new_coors, unq_inv, counts = unique_with_inverse(coors, dims, return_counts=True)
    if mode not in ('max', 'mean', 'sum'):
        raise NotImplementedError
    new_feat = segment_reduce(feat.float(), unq_inv, new_coors.size(0), mode, counts)
    if not return_inv:
        return new_feat, new_coors
    return new_feat, new_coors, unq_inv


def get_activation(activation):
    """sst_ops.py:362-370."""
    if activation == 'relu':
        return torch.nn.functional.relu
    if activation == 'gelu':
        return torch.nn.functional.gelu
    if activation == 'glu':
        return torch.nn.functional.glu
    raise RuntimeError(F'activation should be relu/gelu, not {activation}.')


def get_activation_layer(act, dim=None):
    """sst_ops.py:372-391."""
    act = act.lower()
    if act == 'relu':
        return nn.ReLU(inplace=True)
    if act == 'gelu':
        return nn.GELU()
    if act == 'leakyrelu':
        return nn.LeakyReLU(inplace=True)
    if act == 'prelu':
        return nn.PReLU(num_parameters=dim)
    if act in ('swish', 'silu'):
        return nn.SiLU(inplace=True)
    if act == 'glu':
        return nn.GLU()
    if act == 'elu':
        return nn.ELU(inplace=True)
    raise NotImplementedError


def fuse_norm_act(norm_layer, act_layer):
    """LN followed by exact GELU -> one HIP pass; the activation slot keeps a parameter-free
    Identity so that child indices (state-dict keys) stay those of the reference."""
    if isinstance(norm_layer, LayerNorm) and isinstance(act_layer, nn.GELU):
        norm_layer.fused_act = 'gelu'
        return norm_layer, nn.Identity()
    return norm_layer, act_layer


def build_mlp(in_channel, hidden_dims, norm_cfg, is_head=False, act='relu', bias=False, dropout=0):
    """sst_ops.py:333-360: Sequential of Sequential(Linear, norm, act[, Dropout]) blocks; with
    is_head the last entry is a bare Linear(bias=True).  Keys: <i>.0.weight, <i>.1.{weight,bias}."""
    layer_list = []
    last_channel = in_channel
    if isinstance(hidden_dims, int):
        hidden_dims = [hidden_dims, ]
    for i, c in enumerate(hidden_dims):
        act_layer = get_activation_layer(act, c)
        norm_layer = build_norm_layer(norm_cfg, c)[1]
        if i == len(hidden_dims) - 1 and is_head:
            layer_list.append(nn.Linear(last_channel, c, bias=True))
        else:
            norm_layer, act_layer = fuse_norm_act(norm_layer, act_layer)
            sq = [nn.Linear(last_channel, c, bias=bias), norm_layer, act_layer]
            if dropout > 0:
                sq.append(nn.Dropout(dropout))
            layer_list.append(nn.Sequential(*sq))
        last_channel = c
    return nn.Sequential(*layer_list)
