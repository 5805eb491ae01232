"""Sparse convolution layers -- host mirror of mmdet3d/ops/spconv/conv.py:47-446.

Same constructor arguments, parameter names and weight layout (kD,kH,kW,Cin,Cout)
as the reference, so reference state dicts load; forward keeps the rulebook cache
protocol of SparseConvTensor.indice_dict (conv.py:146-172)."""
import math

import numpy as np
import torch
from torch.nn.parameter import Parameter

from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor
from ..registry import CONV_LAYERS


def _fans_hwio(weight):
    """(fan_in, fan_out) of a [..., Cin, Cout] filter bank: channels times the number of kernel taps."""
    taps = weight.numel() // (weight.shape[-2] * weight.shape[-1])
    return weight.shape[-2] * taps, weight.shape[-1] * taps


def _per_axis(value, ndim):
    return [int(v) for v in value] if isinstance(value, (list, tuple)) else [int(value)] * ndim


class SparseConvolution(SparseModule):
    """Common body of the sparse / sub-manifold / inverse / transposed layers.  Attribute names, the constructor
    signature and the (kD, kH, kW, Cin, Cout) weight layout are the reference's (conv.py:47-111): configs name the
    arguments and state dicts name the parameters."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False):
        super().__init__()
        if groups != 1:
            raise NotImplementedError('grouped sparse convolutions are not built (the reference asserts groups == 1)')
        geom = dict(kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
                    output_padding=output_padding)
        for name, value in geom.items():
            setattr(self, name, _per_axis(value, ndim))
        if any(s != 1 and d != 1 for s, d in zip(self.stride, self.dilation)):
            raise ValueError('stride and dilation cannot both differ from 1 along an axis')
        self.ndim, self.groups = ndim, groups
        self.in_channels, self.out_channels = in_channels, out_channels
        self.subm, self.transposed, self.inverse = subm, transposed, inverse
        self.indice_key, self.fused_bn = indice_key, fused_bn
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.weight = Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        """The reference's initial distribution (conv.py:103-111): Kaiming-uniform(a = sqrt 5) for the weights with
        the fan-in torch derives from a [dim0, dim1, ...] reading of the tensor -- dim 1 times everything behind it,
        which for this channels-last bank is kH * kW * Cin * Cout, not Cin * taps -- and a bias uniform in
        +-1/sqrt(Cin * taps)."""
        fan_w = self.weight[0].numel() if self.weight.dim() > 2 else self.weight.shape[-2]
        bound = math.sqrt(2.0 / (1.0 + 5.0)) * math.sqrt(3.0 / fan_w)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                fan_in, _ = _fans_hwio(self.weight)
                self.bias.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))

    def _ln_wanted(self, indice_pairs, indice_pair_num, num_out):
        from . import modules as _m
        from . import ops as _ops
        kind = _ops.ln_fusion_kind(indice_pairs, indice_pair_num, num_out, self.inverse, self.subm,
                                   self.in_channels, self.out_channels)
        return _m.FUSE_CONV_LN or (kind in ('tile', 'first', 'sorted') and _m.FUSE_TILE_CONV_LN)

    def _ln_fusable(self, features):
        import torch
        from . import ops as _ops
        from .ops import _KD_OK
        return (features.dtype == torch.bfloat16 and self.in_channels in _KD_OK and self.out_channels % 16 == 0
                and self.out_channels in (16, 32, 64, 128) and int(np.prod(self.kernel_size)) <= 32
                and features.shape[0] > 0)

    def forward(self, input, _ln=None):
        """``_ln``: the LayerNorm module that follows this conv in a make_sparse_convmodule block; when the
        shape allows, it (and its fused GELU) run in the conv kernel's epilogue and the returned tensor is
        marked ``_ln_applied`` so that the container skips the norm."""
        if not isinstance(input, SparseConvTensor):
            raise TypeError(f'{type(self).__name__} takes a SparseConvTensor, got {type(input).__name__}')
        features, batch_size = input.features, input.batch_size

        def wrap(feats, ids, shape):
            out = SparseConvTensor(feats, ids, shape, batch_size, input.grid)
            out.indice_dict = input.indice_dict
            return out

        if self.conv1x1:  # a 1^ndim kernel is a per-voxel Linear: no rulebook at all
            feats = torch.mm(features, self.weight.view(self.in_channels, self.out_channels).to(features.dtype))
            if self.bias is not None:
                feats = feats + self.bias.to(feats.dtype)
            return wrap(feats, input.indices, input.spatial_shape)

        # the rulebook: shared through indice_dict between layers that name the same indice_key
        cached = input.find_indice_pair(self.indice_key)
        if self.inverse:
            if cached is None:
                raise KeyError(f'inverse conv: no rulebook cached under indice_key {self.indice_key!r}')
            _, outids, indice_pairs, indice_pair_num, out_spatial_shape = cached  # the partner's INPUT side
            if indice_pairs.shape[0] != int(np.prod(self.kernel_size)):
                raise ValueError('an inverse conv needs the kernel size of the conv it undoes')
        else:
            if self.subm:
                out_spatial_shape = input.spatial_shape
            elif self.transposed:
                out_spatial_shape = ops.get_deconv_output_size(input.spatial_shape, self.kernel_size, self.stride,
                                                               self.padding, self.dilation, self.output_padding)
            else:
                out_spatial_shape = ops.get_conv_output_size(input.spatial_shape, self.kernel_size, self.stride,
                                                             self.padding, self.dilation)
            if cached is not None:
                outids, _, indice_pairs, indice_pair_num, _ = cached
            else:
                outids, indice_pairs, indice_pair_num = ops.get_indice_pairs(
                    input.indices, batch_size, input.spatial_shape, self.kernel_size, self.stride, self.padding,
                    self.dilation, self.output_padding, self.subm, self.transposed, grid=input.grid)
                input.indice_dict[self.indice_key] = (outids, input.indices, indice_pairs, indice_pair_num,
                                                      input.spatial_shape)
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
                out_tensor = wrap(out_features, outids, out_spatial_shape)
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
        return wrap(out_features, outids, out_spatial_shape)


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
class SparseConvTranspose2d(SparseConvolution):
    """conv.py:286-310 of the reference: output sites by the transposed rule (get_deconv_output_size)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, transposed=True, indice_key=indice_key)


@CONV_LAYERS.register_module()
class SparseConvTranspose3d(SparseConvolution):
    """conv.py:313-337"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, transposed=True, indice_key=indice_key)


@CONV_LAYERS.register_module()
class SparseInverseConv2d(SparseConvolution):
    """conv.py:340-356"""

    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(2, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)


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
