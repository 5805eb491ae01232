from .scatter_points import (DynamicScatter, dynamic_scatter, gather_rows, object_grid_geometry, segment_reduce,
                             voxelize_scatter_mean)
from .voxelize import Voxelization, voxelization

__all__ = ['Voxelization', 'voxelization', 'dynamic_scatter', 'DynamicScatter', 'segment_reduce', 'gather_rows',
           'voxelize_scatter_mean', 'object_grid_geometry']
