"""SparseConvTensor -- same fields and methods as mmdet3d/ops/spconv/structure.py:21-69."""
import numpy as np
import torch


def scatter_nd(indices, updates, shape):
    """structure.py:5-18: dense tensor with `updates` written at `indices` (no repeats)."""
    ret = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    ndim = indices.shape[-1]
    output_shape = list(indices.shape[:-1]) + shape[indices.shape[-1]:]
    flat = indices.view(-1, ndim)
    slices = [flat[:, i] for i in range(ndim)] + [Ellipsis]
    ret[slices] = updates.view(*output_shape)
    return ret


class SparseConvTensor(object):

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        output_shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        res = scatter_nd(self.indices.long(), self.features, output_shape)
        if not channels_first:
            return res
        ndim = len(self.spatial_shape)
        trans = list(range(0, ndim + 1))
        trans.insert(1, ndim + 1)
        return res.permute(*trans).contiguous()

    def replace_feature(self, new_features):
        """spconv 2.x style functional update (sparse_block.py:13-19 probes for it)."""
        out = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size,
                               self.grid)
        out.indice_dict = self.indice_dict
        return out

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
