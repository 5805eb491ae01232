from .scatter_points import DynamicScatter, dynamic_scatter, segment_reduce
from .voxelize import Voxelization, voxelization

__all__ = ['Voxelization', 'voxelization', 'dynamic_scatter', 'DynamicScatter', 'segment_reduce']
