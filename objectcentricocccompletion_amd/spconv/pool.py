"""Sparse max pooling -- host mirror of mmdet3d/ops/spconv/pool.py (SparseMaxPool :20-73, SparseMaxPool2d / 3d :76-87).
The pooling itself: spconv.ops.indice_maxpool (csrc/sparse_pool.hip), with the reference's zero-initialised output."""
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


class SparseMaxPool(SparseModule):

    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        per_axis = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        self.ndim = ndim
        self.kernel_size, self.stride = per_axis(kernel_size), per_axis(stride)
        self.padding, self.dilation = per_axis(padding), per_axis(dilation)
        self.subm = subm

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features, indices = input.features, input.indices
        spatial_shape, batch_size = input.spatial_shape, input.batch_size
        if not self.subm:
            out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding,
                                                         self.dilation)
        else:
            out_spatial_shape = spatial_shape
        outids, indice_pairs, indice_pairs_num = ops.get_indice_pairs(indices, batch_size, spatial_shape, self.kernel_size,
                                                                      self.stride, self.padding, self.dilation, 0, self.subm)
        out_features = Fsp.indice_maxpool(features, indice_pairs, indice_pairs_num, outids.shape[0])
        out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseMaxPool2d(SparseMaxPool):

    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(2, kernel_size, stride, padding, dilation)


class SparseMaxPool3d(SparseMaxPool):

    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(3, kernel_size, stride, padding, dilation)
