"""SparseConvTensor: the active voxels of a batch of grids.  Field and method names follow the reference's
container (mmdet3d/ops/spconv/structure.py:21-69) because configs, blocks and checkpoints address them."""
import math

import torch


def scatter_nd(indices, updates, shape):
    """Dense tensor of `shape` holding `updates` at the integer coordinates `indices` [..., k] (k leading dims of
    `shape`; coordinates must not repeat), zero elsewhere.  One index_put."""
    k = indices.shape[-1]
    dense = updates.new_zeros(tuple(shape))
    coords = indices.reshape(-1, k).long().unbind(1)
    dense.index_put_(coords, updates.reshape((-1,) + tuple(shape[k:])))
    return dense


class SparseConvTensor(object):
    """features [N, C]; indices [N, 1 + ndim] int32 rows (batch, z, y, x); the grid extent; and ``indice_dict``, the
    rulebooks already built on this geometry, keyed by the layers' ``indice_key``.  ``grid`` is kept for
    call-compatibility with the reference (its pre-allocated dense index grid); the product never needs one."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices.int() if indices.dtype != torch.int32 else indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.grid = grid
        self.indice_dict = {}

    def find_indice_pair(self, key):
        return self.indice_dict.get(key) if key is not None else None

    @property
    def spatial_size(self):
        return math.prod(int(s) for s in self.spatial_shape)

    @property
    def sparity(self):
        """Fraction of the batch's cells that are active (the reference spells it this way)."""
        return self.indices.shape[0] / (self.spatial_size * self.batch_size)

    def dense(self, channels_first=True):
        """[B, C, *spatial] (or [B, *spatial, C]) with zeros at inactive cells."""
        extent = (int(self.batch_size),) + tuple(int(s) for s in self.spatial_shape)
        out = scatter_nd(self.indices, self.features, extent + (self.features.shape[1],))
        return out.movedim(-1, 1).contiguous() if channels_first else out

    def replace_feature(self, new_features):
        """spconv 2.x style functional update (sparse_block.py:13-19 probes for it): same geometry, same rulebooks."""
        twin = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.grid)
        twin.indice_dict = self.indice_dict
        return twin
