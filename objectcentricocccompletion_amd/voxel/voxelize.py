"""Voxelisation ops -- host mirror of mmdet3d/ops/voxel/voxelize.py:10-113.

Same names, arguments and return values as the reference; the arithmetic runs
in the HIP kernels ococc_dynamic_voxelize_f32 / ococc_hard_voxelize_f32.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from .. import _lib as L


def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """In-place variant with the signature of voxel_layer.dynamic_voxelize
    (mmdet3d/ops/voxel/src/voxelization.h:77-88): fills coors [N,3] int32 (z,y,x)."""
    assert NDim == 3
    L.require_device(points, coors)
    assert points.dtype == torch.float32 and coors.dtype == torch.int32
    points = points.contiguous()
    assert coors.is_contiguous()
    L.check(L.lib.ococc_dynamic_voxelize_f32(L.ptr(points), points.size(0), points.size(1),
                                             L.f3(voxel_size), L.f6(coors_range), L.ptr(coors),
                                             L.stream()), 'dynamic_voxelize')


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
                  max_points, max_voxels, NDim=3):
    """voxel_layer.hard_voxelize (voxelization.h:60-75): fills the caller's buffers and
    returns the number of voxels (a host int, like the reference -- one device sync)."""
    assert NDim == 3
    L.require_device(points, voxels, coors, num_points_per_voxel)
    points = points.contiguous()
    vs, rng = L.f3(voxel_size), L.f6(coors_range)
    nbytes = L.lib.ococc_hard_voxelize_workspace_bytes(points.size(0), vs, rng)
    if nbytes < 0:
        raise L.OcoccError('hard_voxelize: bad voxel_size / coors_range')
    ws = L.workspace(nbytes, points.device)
    vn = torch.zeros(1, dtype=torch.int32, device=points.device)
    L.check(L.lib.ococc_hard_voxelize_f32(L.ptr(points), points.size(0), points.size(1), vs, rng,
                                          int(max_points), int(max_voxels), L.ptr(voxels),
                                          L.ptr(coors), L.ptr(num_points_per_voxel), L.ptr(vn),
                                          L.ptr(ws), ws.numel(), L.stream()), 'hard_voxelize')
    return int(vn.item())


class _Voxelization(Function):

    @staticmethod
    def forward(ctx, points, voxel_size, coors_range, max_points=35, max_voxels=20000):
        """points [N, >=3] -> coors [N,3] (dynamic, max_points == -1 or max_voxels == -1) or
        (voxels [M,max_points,ndim], coors [M,3], num_points_per_voxel [M])."""
        if max_points == -1 or max_voxels == -1:
            coors = points.new_empty(size=(points.size(0), 3), dtype=torch.int)  # every row is written
            dynamic_voxelize(points, coors, voxel_size, coors_range, 3)
            return coors
        voxels = points.new_zeros(size=(max_voxels, max_points, points.size(1)))
        coors = points.new_zeros(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = points.new_zeros(size=(max_voxels, ), dtype=torch.int)
        voxel_num = hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size,
                                  coors_range, max_points, max_voxels, 3)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    """Same constructor and forward as mmdet3d/ops/voxel/voxelize.py:66-113."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid_size
        self.pcd_shape = [*grid_size[:2], 1][::-1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points,
                            max_voxels)

    def __repr__(self):
        return (f'{self.__class__.__name__}(voxel_size={self.voxel_size}, '
                f'point_cloud_range={self.point_cloud_range}, '
                f'max_num_points={self.max_num_points}, max_voxels={self.max_voxels})')
