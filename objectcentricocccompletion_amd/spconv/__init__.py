"""Sparse convolution -- host mirror of mmdet3d/ops/spconv (vendored spconv 1.x API,
what SpConv2 replaces at run time through overwrite_spconv/write_spconv2.py)."""
from .conv import (SparseConv2d, SparseConv3d, SparseConvolution, SparseConvTranspose2d, SparseConvTranspose3d,
                   SparseInverseConv2d, SparseInverseConv3d, SubMConv2d, SubMConv3d)
from .modules import RemoveGrid, SparseModule, SparseSequential, ToDense
from .pool import SparseMaxPool2d, SparseMaxPool3d
from .structure import SparseConvTensor, scatter_nd

__all__ = ['SparseConvolution', 'SparseConv2d', 'SparseConv3d', 'SubMConv2d', 'SubMConv3d',
           'SparseInverseConv3d', 'SparseInverseConv2d', 'SparseConvTranspose2d', 'SparseConvTranspose3d', 'SparseModule', 'SparseSequential', 'SparseConvTensor',
           'scatter_nd', 'ToDense', 'SparseMaxPool2d', 'SparseMaxPool3d', 'RemoveGrid']
