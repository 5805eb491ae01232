"""SubMConv3d-only occupancy encoder over per-object voxel grids -- the workload
BASELINE.json configs[1] names (64 object grids of 40^3 cells at 0.2 m).

Built only from the reference's own operators, in the order its voxel pipelines use them:
Voxelization (dynamic, mmdet3d/ops/voxel/voxelize.py:10-113) -> DynamicScatter mean
(mmdet3d/ops/voxel/scatter_points.py:53-107) -> SparseConvTensor -> a stack of
make_sparse_convmodule(SubMConv3d 3x3x3 -> LN(eps 1e-3) -> GELU) sharing one indice_key
(mmdet3d/ops/sparse_block.py:216-289), channels 16 -> 32 -> 64 -> 128.
"""
import torch
from torch import nn

from .sparse_block import make_sparse_convmodule
from .spconv import SparseConvTensor
from .spconv import ops as sp_ops
from .spconv.functional import chain_ln_backward
from .voxel import dynamic_scatter, object_grid_geometry, voxelization, voxelize_scatter_mean


class SubMOccEncoder(nn.Module):

    def __init__(self, in_channels=16, channels=(32, 64, 128), voxel_size=(0.2, 0.2, 0.2),
                 point_cloud_range=(-4, -4, -4, 4, 4, 4), norm_cfg=dict(type='LN', eps=1e-3),
                 act_type='gelu', feature_dtype=torch.bfloat16, fused_front_end=True, grouped_points=False):
        """``grouped_points=True`` is the caller's promise that the points of one object grid are contiguous in the
        input (batch_idx non-decreasing), as the per-object pipelines and synthetic_object_grids deliver them; the
        fixed-capacity geometry then runs through the per-grid LDS kernels (voxel.object_grid_geometry).  The
        reference's voxelisation has no such requirement, so the default is the general path.  With the promise
        given, a batch that breaks it is reported through the status word of the returned meta tensor (its stray
        points are dropped); ``check_geometry`` reads that word back and raises -- call it outside graph capture."""
        super().__init__()
        self.grouped_points = bool(grouped_points)
        self.fused_front_end = bool(fused_front_end) and feature_dtype in (torch.bfloat16, torch.float32)
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = list(point_cloud_range)
        self.grid = [int(round((point_cloud_range[3 + i] - point_cloud_range[i]) / voxel_size[i]))
                     for i in range(3)]  # x, y, z cells
        self.sparse_shape = self.grid[::-1]  # (D, H, W) = (z, y, x)
        self.feature_dtype = feature_dtype
        layers = []
        c = in_channels
        for co in channels:
            layers.append(make_sparse_convmodule(c, co, 3, 'subm1', padding=1,
                                                 conv_type='SubMConv3d', act_type=act_type,
                                                 norm_cfg=norm_cfg))
            c = co
        self.conv_layers = nn.ModuleList(layers)

    def voxelize(self, points, batch_idx, batch_size):
        """points [N, 3+] f32 in the object frame, batch_idx [N] int32 -> voxel features,
        voxel coordinates (b,z,y,x)."""
        zyx = voxelization(points, self.voxel_size, self.point_cloud_range, -1, -1)
        coors = torch.cat([batch_idx.view(-1, 1).to(torch.int32), zyx], 1)
        return coors

    def geometry(self, points, feats, batch_idx, batch_size, static=False):
        """Everything that depends on the points only, none of it on the weights: voxelise -> scatter-mean ->
        SparseConvTensor -> the sub-manifold rulebook all conv layers share (indice_key 'subm1').  A training loop
        can run this for batch t+1 on a second stream while batch t trains (graph.PipelinedStep): the chain is a
        dozen short, latency-bound launches that fit beside the convolutions.  Returns the input SparseConvTensor
        with the rulebook already in its indice_dict; pass it to forward(geometry=...)."""
        key = self.conv_layers[0][0].indice_key
        conv = self.conv_layers[0][0]
        if (static and self.grouped_points and self.fused_front_end and list(conv.kernel_size) == [3, 3, 3]
                and list(conv.dilation) == [1, 1, 1] and not (feats.requires_grad and torch.is_grad_enabled())):
            res = object_grid_geometry(points, batch_idx, feats, self.voxel_size, self.point_cloud_range,
                                       self.sparse_shape, batch_size, out_dtype=self.feature_dtype)
            if res is not None:
                vfeats, vcoors, _, _, meta, pairs, num = res
                x = SparseConvTensor(vfeats, vcoors, self.sparse_shape, batch_size)
                x.indice_dict[key] = (vcoors, vcoors, pairs, num, self.sparse_shape)
                x.meta = meta
                return x
        x = self._front_end(points, feats, batch_idx, batch_size, static)
        outids, pairs, num = sp_ops.get_indice_pairs(x.indices, batch_size, self.sparse_shape, conv.kernel_size,
                                                     conv.stride, conv.padding, conv.dilation, conv.output_padding,
                                                     True, False, grid=x.grid)
        x.indice_dict[key] = (outids, x.indices, pairs, num, self.sparse_shape)
        return x

    @staticmethod
    def check_geometry(x):
        """Read back the status word of a grouped-points geometry (one device sync; not inside a graph capture) and
        raise if the batch broke the ordering promise or overflowed its fixed capacity."""
        meta = getattr(x, 'meta', None)
        if meta is not None and int(meta[1]) != 0:
            raise ValueError('object_grid_geometry: batch_idx is not non-decreasing (or a grid overflowed its capacity); '
                             'stray points were dropped -- build the encoder with grouped_points=False for such batches')
        return x

    def prepare_weights(self, grad=None):
        """All conv weights to their bf16 kernel layouts in one launch (forward operands, and the dgrad operands of the
        layers whose input needs a gradient).  forward() does this itself unless told the operands are current."""
        grad = torch.is_grad_enabled() if grad is None else grad
        items = [(layer[0].weight, 0) for layer in self.conv_layers]
        items += [(layer[0].weight, 1) for layer in self.conv_layers[1:]] if grad else []
        sp_ops.prepare_weights(items)

    def forward(self, points=None, feats=None, batch_idx=None, batch_size=None, static=False, geometry=None,
                weights_ready=False):
        """``static=True`` keeps every tensor at its fixed capacity (one row per point; unused
        voxel rows carry -1 coordinates, take part in no rulebook pair and must get a zero
        upstream gradient), so that the whole step has no device read-back and can be captured
        in a HIP graph (graph.GraphedStep).  ``geometry``: the result of self.geometry() for this batch."""
        # ``weights_ready``: prepare_weights() already ran for the current parameter values (a pipelined step does it
        # behind the optimizer, beside the next batch's geometry)
        if not weights_ready:
            self.prepare_weights()
        x = geometry if geometry is not None else self.geometry(points, feats, batch_idx, batch_size, static)
        # (a plain stack: every block's output goes into the next block's convolution and nowhere else, so each
        # LayerNorm backward may run inside the next layer's input-gradient kernel)
        with chain_ln_backward():
            for layer in self.conv_layers:
                x = layer(x)
        return x

    def _front_end(self, points, feats, batch_idx, batch_size, static):
        if self.fused_front_end:
            # voxelize -> cat -> DynamicScatter(mean) -> cast in one C-ABI call (7 launches instead of 15)
            vfeats, vcoors, _, _, _ = voxelize_scatter_mean(
                points, batch_idx, feats, self.voxel_size, self.point_cloud_range, self.sparse_shape, batch_size,
                static=static, out_dtype=self.feature_dtype)
        else:
            coors = self.voxelize(points, batch_idx, batch_size)
            vfeats, vcoors = dynamic_scatter(feats, coors, 'mean',
                                             grid_shape=[batch_size] + self.sparse_shape, static=static)
            vfeats = vfeats.to(self.feature_dtype)
        return SparseConvTensor(vfeats, vcoors, self.sparse_shape, batch_size)


def synthetic_object_grids(num_grids, points_per_grid, in_channels=16, half_extent=4.0, seed=0,
                           device='cuda'):
    """Random-point object grids of the benchmark shape: points uniform in the
    [-4, 4)^3 m object box (40^3 cells at 0.2 m), features = xyz, two attributes
    (intensity, elongation), offset to the voxel centre, zero padded to in_channels."""
    g = torch.Generator().manual_seed(seed)
    n = num_grids * points_per_grid
    xyz = (torch.rand(n, 3, generator=g) * 2 - 1) * half_extent
    attr = torch.rand(n, 2, generator=g)
    centre = (torch.floor((xyz + half_extent) / 0.2) + 0.5) * 0.2 - half_extent
    feats = torch.zeros(n, in_channels)
    feats[:, :3] = xyz
    feats[:, 3:5] = attr
    feats[:, 5:8] = xyz - centre
    batch_idx = torch.arange(num_grids, dtype=torch.int32).repeat_interleave(points_per_grid)
    return xyz.to(device), feats.to(device), batch_idx.to(device)
