"""SparseModule / SparseSequential / ToDense -- host mirror of
mmdet3d/ops/spconv/modules.py:44-214 (container logic only; no device code)."""
import os
from collections import OrderedDict

import torch
from torch import nn

from .structure import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: SparseSequential hands such modules the SparseConvTensor itself."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


# Run the LayerNorm (+GELU) that follows a sparse conv inside the conv kernel's epilogue
# (ococc_sparse_conv_gather_gemm_ln_bf16).  Off by default: measured on configs[1] the fused epilogue costs
# what the separate LN kernel costs (the workgroups of the conv kernel reach their epilogue together, so the
# extra erf / statistics work is not hidden behind anyone's MFMA phase) -- 560 vs 551 us of kernels per step.
FUSE_CONV_LN = os.environ.get('OCOCC_FUSE_CONV_LN', '0') == '1'
# The tile kernel (csrc/sparse_conv_tile.hip) is the exception: its epilogue has the whole f32 row in LDS, the
# statistics are three lane exchanges, and the layer's separate LN launch (10.8 us on the 32 -> 64 layer of
# configs[1]) disappears.  On by default for the layers that run on that kernel, and for the 16 -> 32 input layer
# (resident-weights kernel, 19.9 us fused against 15.5 + 7.4 us).
FUSE_TILE_CONV_LN = os.environ.get('OCOCC_FUSE_TILE_CONV_LN', '1') == '1'


def is_sparse_conv(module):
    """modules.py:27-29"""
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


def _mean_update(vals, m_vals, t):
    """running mean of one value or a list of values after t earlier updates (modules.py:32-43)"""
    single = not isinstance(vals, list)
    vals = [vals] if single else vals
    m_vals = m_vals if isinstance(m_vals, list) else [m_vals]
    outputs = [t / float(t + 1) * m_val + 1 / float(t + 1) * val for val, m_val in zip(vals, m_vals)]
    return outputs[0] if len(outputs) == 1 else outputs


def _is_fusable_norm(module):
    from ..norm import LayerNorm
    return (FUSE_CONV_LN or FUSE_TILE_CONV_LN) and isinstance(module, LayerNorm) \
        and module.fused_act in ('none', 'gelu')


class SparseSequential(SparseModule):
    """Sequential container: sparse modules receive the SparseConvTensor, dense ones
    (norm, activation) its feature matrix (modules.py:127-140)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        named = args[0].items() if len(args) == 1 and isinstance(args[0], OrderedDict) else \
            ((str(i), m) for i, m in enumerate(args))
        for name, module in list(named) + list(kwargs.items()):
            self.add(module, name)
        self._sparity_dict = {}

    def add(self, module, name=None):
        """Append a child; unnamed children are numbered like nn.Sequential's."""
        name = str(len(self._modules)) if name is None else name
        if name in self._modules:
            raise KeyError(f'a child named {name!r} exists')
        self.add_module(name, module)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        children = list(self._modules.values())
        if not -len(children) <= idx < len(children):
            raise IndexError(f'index {idx} is out of range')
        return children[idx]

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def forward(self, input):
        mods = list(self._modules.items())
        skip = False
        for i, (k, module) in enumerate(mods):
            if skip:  # this norm already ran inside the preceding conv kernel
                skip = False
                continue
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                nxt = mods[i + 1][1] if i + 1 < len(mods) else None
                if _is_fusable_norm(nxt) and hasattr(module, '_ln_fusable'):
                    input = module(input, _ln=nxt)
                    if getattr(input, '_ln_applied', False):
                        input._ln_applied = False
                        skip = True
                else:
                    input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input.features = module(input.features)
            else:
                input = module(input)
        return input


    def fused(self):
        """A copy of the container with every (sparse conv, BatchNorm1d) pair folded into one conv with a bias, for
        inference (modules.py:139-185; the reference marks it "don't use this").  Same folding arithmetic as there --
        including its ``sqrt(running_var) + eps`` denominator, which is not BatchNorm's ``sqrt(running_var + eps)``."""
        from .conv import SparseConvolution
        mods = list(self._modules.values())
        fused_mods, idx = [], 0
        while idx < len(mods):
            if is_sparse_conv(mods[idx]) and idx < len(mods) - 1 and isinstance(mods[idx + 1], nn.BatchNorm1d):
                old, bn = mods[idx], mods[idx + 1]
                conv = SparseConvolution(ndim=old.ndim, in_channels=old.in_channels, out_channels=old.out_channels,
                                         kernel_size=old.kernel_size, stride=old.stride, padding=old.padding,
                                         dilation=old.dilation, groups=old.groups, bias=True, subm=old.subm,
                                         output_padding=old.output_padding, transposed=old.transposed,
                                         inverse=old.inverse, indice_key=old.indice_key, fused_bn=True)
                conv.load_state_dict(old.state_dict(), False)
                conv.to(old.weight.device)
                with torch.no_grad():
                    scale = bn.weight / (torch.sqrt(bn.running_var) + bn.eps)
                    conv.bias.zero_()
                    conv.weight.mul_(scale)
                    conv.bias.copy_((conv.bias - bn.running_mean) * scale + bn.bias)
                fused_mods.append(conv)
                idx += 2
            else:
                fused_mods.append(mods[idx])
                idx += 1
        return SparseSequential(*fused_mods)


class RemoveGrid(SparseModule):
    """drops the pre-allocated grid buffer of the tensor passing through (modules.py:197-202)"""

    def forward(self, x: SparseConvTensor):
        x.grid = None
        return x


class ToDense(SparseModule):
    """SparseConvTensor -> dense N C D H W tensor (modules.py:200-204)."""

    def forward(self, x: SparseConvTensor):
        return x.dense()
