"""Sparse convolution layers -- host mirror of mmdet3d/ops/spconv/conv.py:47-446.

Same constructor arguments, parameter names and weight layout (kD,kH,kW,Cin,Cout)
as the reference, so reference state dicts load; forward keeps the rulebook cache
protocol of SparseConvTensor.indice_dict (conv.py:146-172)."""
import math

import numpy as np
import torch
from torch.nn import init
from torch.nn.parameter import Parameter

from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor
from ..registry import CONV_LAYERS


def _calculate_fan_in_and_fan_out_hwio(tensor):
    """conv.py:28-45 (weights are stored ...,Cin,Cout)."""
    dimensions = tensor.ndimension()
    if dimensions < 2:
        raise ValueError('fan in and fan out can not be computed for tensor with fewer than 2 dimensions')
    if dimensions == 2:
        return tensor.size(-2), tensor.size(-1)
    receptive = tensor[..., 0, 0].numel()
    return tensor.size(-2) * receptive, tensor.size(-1) * receptive


class SparseConvolution(SparseModule):

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False):
        super().__init__()
        assert groups == 1

        def _l(v):
            return list(v) if isinstance(v, (list, tuple)) else [v] * ndim

        kernel_size, stride, padding = _l(kernel_size), _l(stride), _l(padding)
        dilation, output_padding = _l(dilation), _l(output_padding)
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.conv1x1 = np.prod(kernel_size) == 1
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = output_padding
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = _calculate_fan_in_and_fan_out_hwio(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _ln_wanted(self, indice_pairs, indice_pair_num, num_out):
        from . import modules as _m
        from . import ops as _ops
        kind = _ops.ln_fusion_kind(indice_pairs, indice_pair_num, num_out, self.inverse, self.subm,
                                   self.in_channels, self.out_channels)
        return _m.FUSE_CONV_LN or (kind in ('tile', 'first') and _m.FUSE_TILE_CONV_LN)

    def _ln_fusable(self, features):
        import torch
        from . import ops as _ops
        from .ops import _KD_OK
        return (_ops._probe is None and features.dtype == torch.bfloat16 and self.in_channels in _KD_OK and self.out_channels % 16 == 0
                and self.out_channels in (16, 32, 64, 128) and int(np.prod(self.kernel_size)) <= 32
                and features.shape[0] > 0)

    def forward(self, input, _ln=None):
        """``_ln``: the LayerNorm module that follows this conv in a make_sparse_convmodule block; when the
        shape allows, it (and its fused GELU) run in the conv kernel's epilogue and the returned tensor is
        marked ``_ln_applied`` so that the container skips the norm."""
        assert isinstance(input, SparseConvTensor)
        features = input.features
        indices = input.indices
        spatial_shape = input.spatial_shape
        batch_size = input.batch_size
        if not self.subm:
            if self.transposed:
                out_spatial_shape = ops.get_deconv_output_size(
                    spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation,
                    self.output_padding)
            else:
                out_spatial_shape = ops.get_conv_output_size(
                    spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation)
        else:
            out_spatial_shape = spatial_shape
        if self.conv1x1:
            features = torch.mm(input.features,
                                self.weight.view(self.in_channels, self.out_channels).to(features.dtype))
            if self.bias is not None:
                features = features + self.bias.to(features.dtype)
            out_tensor = SparseConvTensor(features, input.indices, input.spatial_shape,
                                          input.batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            return out_tensor
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None
            _, outids, indice_pairs, indice_pair_num, out_spatial_shape = datas
            assert indice_pairs.shape[0] == np.prod(self.kernel_size), \
                'inverse conv must have same kernel size as its couple conv'
        else:
            if self.indice_key is not None and datas is not None:
                outids, _, indice_pairs, indice_pair_num, _ = datas
            else:
                outids, indice_pairs, indice_pair_num = ops.get_indice_pairs(
                    indices, batch_size, spatial_shape, self.kernel_size, self.stride,
                    self.padding, self.dilation, self.output_padding, self.subm, self.transposed,
                    grid=input.grid)
                input.indice_dict[self.indice_key] = (outids, indices, indice_pairs,
                                                      indice_pair_num, spatial_shape)
        if self.fused_bn:
            assert self.bias is not None
            out_features = ops.fused_indice_conv(features, self.weight, self.bias, indice_pairs,
                                                 indice_pair_num, outids.shape[0], self.inverse,
                                                 self.subm)
        else:
            if _ln is not None and self.bias is None and self._ln_fusable(features) and self._ln_wanted(
                    indice_pairs, indice_pair_num, outids.shape[0]):
                # norm (+ GELU) of the enclosing make_sparse_convmodule, fused into the conv epilogue
                out_features = Fsp.indice_conv_ln(features, self.weight, _ln.weight, _ln.bias, indice_pairs,
                                                  indice_pair_num, outids.shape[0], _ln.eps,
                                                  1 if _ln.fused_act == 'gelu' else 0, self.inverse, self.subm)
                out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
                out_tensor.indice_dict = input.indice_dict
                out_tensor.grid = input.grid
                out_tensor._ln_applied = True
                return out_tensor
            if self.subm:
                out_features = Fsp.indice_subm_conv(features, self.weight, indice_pairs,
                                                    indice_pair_num, outids.shape[0])
            elif self.inverse:
                out_features = Fsp.indice_inverse_conv(features, self.weight, indice_pairs,
                                                       indice_pair_num, outids.shape[0])
            else:
                out_features = Fsp.indice_conv(features, self.weight, indice_pairs,
                                               indice_pair_num, outids.shape[0])
            if self.bias is not None:
                out_features = out_features + self.bias.to(out_features.dtype)
        out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


@CONV_LAYERS.register_module()
class SparseConv2d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, indice_key=indice_key)


@CONV_LAYERS.register_module()
class SparseConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, indice_key=indice_key)


@CONV_LAYERS.register_module()
class SparseInverseConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)


@CONV_LAYERS.register_module()
class SubMConv2d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, True, indice_key=indice_key)


@CONV_LAYERS.register_module()
class SubMConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, True, indice_key=indice_key)
