"""Sparse max pooling modules -- the API of mmdet3d/ops/spconv/pool.py (SparseMaxPool :20-73, SparseMaxPool2d / 3d
:76-87).  The pooling itself is spconv.ops.indice_maxpool (csrc/sparse_pool.hip), with the reference's zero-initialised
output; the rulebook is the regular (or sub-manifold) one of the convolutions, never cached under an indice_key."""
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


def _axes(value, ndim):
    return list(value) if isinstance(value, (list, tuple)) else [value] * ndim


class SparseMaxPool(SparseModule):
    """``ndim``-dimensional window maximum over the active sites; ``subm=True`` keeps the input's sites."""

    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        self.ndim, self.subm = ndim, subm
        self.kernel_size, self.stride = _axes(kernel_size, ndim), _axes(stride, ndim)
        self.padding, self.dilation = _axes(padding, ndim), _axes(dilation, ndim)

    def output_shape(self, spatial_shape):
        if self.subm:
            return spatial_shape
        return ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation)

    def forward(self, input):
        if not isinstance(input, SparseConvTensor):
            raise TypeError(f'{type(self).__name__} takes a SparseConvTensor, got {type(input).__name__}')
        out_ids, pairs, pair_num = ops.get_indice_pairs(input.indices, input.batch_size, input.spatial_shape,
                                                        self.kernel_size, self.stride, self.padding, self.dilation, 0,
                                                        self.subm)
        pooled = Fsp.indice_maxpool(input.features, pairs, pair_num, out_ids.shape[0])
        result = SparseConvTensor(pooled, out_ids, self.output_shape(input.spatial_shape), input.batch_size, input.grid)
        result.indice_dict = input.indice_dict
        return result


def _fixed_ndim(ndim):
    class _Pool(SparseMaxPool):
        def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
            super().__init__(ndim, kernel_size, stride, padding, dilation)
    _Pool.__name__ = _Pool.__qualname__ = f'SparseMaxPool{ndim}d'
    return _Pool


SparseMaxPool2d = _fixed_ndim(2)
SparseMaxPool3d = _fixed_ndim(3)
