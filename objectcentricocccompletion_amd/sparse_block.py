"""Sparse conv building blocks -- host mirror of mmdet3d/ops/sparse_block.py
(make_sparse_convmodule :216-289, SparseBasicBlock :81-144).

The module tree (child names '0','1','2' of the SparseSequential; conv1/norm1/...) and
parameter names equal the reference's, so its checkpoints load.  Where the reference
order is conv -> LN -> GELU the GELU is folded into the LayerNorm kernel and the
activation slot holds nn.Identity (parameter free, state dict unchanged)."""
from torch import nn

from .norm import LayerNorm
from .registry import build_conv_layer, build_norm_layer
from .spconv import SparseModule, SparseSequential


def replace_feature(out, new_features):
    """sparse_block.py:13-19."""
    if 'replace_feature' in out.__dir__():
        return out.replace_feature(new_features)
    out.features = new_features
    return out


def _act_layer(act_type):
    act_type = act_type.lower()
    if act_type == 'relu':
        return nn.ReLU(inplace=True)
    if act_type == 'gelu':
        return nn.GELU()
    if act_type == 'silu':
        return nn.SiLU(inplace=True)
    raise NotImplementedError


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type='SubMConv3d', act_type='relu', norm_cfg=None,
                           order=('conv', 'norm', 'act')):
    """Same signature and result structure as sparse_block.py:216-289."""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {'conv', 'norm', 'act'} == {'conv', 'norm', 'act'}
    conv_cfg = dict(type=conv_type, indice_key=indice_key)
    layers = list()
    for layer in order:
        if layer == 'conv':
            if conv_type not in ['SparseInverseConv4d', 'SparseInverseConv3d',
                                 'SparseInverseConv2d', 'SparseInverseConv1d']:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size,
                                               stride=stride, padding=padding, bias=False))
            else:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size,
                                               bias=False))
        elif layer == 'norm':
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == 'act':
            layers.append(_act_layer(act_type))
    # fold  LN -> GELU  into one kernel
    for i in range(len(layers) - 1):
        if isinstance(layers[i], LayerNorm) and isinstance(layers[i + 1], nn.GELU):
            layers[i].fused_act = 'gelu'
            layers[i + 1] = nn.Identity()
    return SparseSequential(*layers)


class SparseBasicBlock(SparseModule):
    """Residual block of two sub-manifold convs (sparse_block.py:81-144; the reference
    inherits the layer names conv1/norm1(bn1)/conv2/norm2(bn2)/relu from mmdet's BasicBlock)."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None,
                 act_type='relu'):
        super().__init__()
        assert conv_cfg is not None and norm_cfg is not None
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=1,
                                      bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = _act_layer(act_type)
        self.downsample = downsample
        self.stride = stride

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    def forward(self, x):
        identity = x.features
        assert x.features.dim() == 2, f'x.features.dim()={x.features.dim()}'
        out = self.conv1(x)
        out = replace_feature(out, self.norm1(out.features))
        out = replace_feature(out, self.relu(out.features))
        out = self.conv2(out)
        out = replace_feature(out, self.norm2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = replace_feature(out, out.features + identity)
        out = replace_feature(out, self.relu(out.features))
        return out


class SparseBottleneck(SparseModule):
    """Bottleneck block of sub-manifold convs (sparse_block.py:22-78; the reference inherits the constructor and the
    layer names conv1 / bn1 / conv2 / bn2 / conv3 / bn3 / relu / downsample from mmdet's ResNet Bottleneck, 'pytorch'
    style: 1 x 1 -> 3 x 3 (the stride sits here) -> 1 x 1 onto ``planes * 4`` channels).  Not used by ococcnet.py."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        assert conv_cfg is not None and norm_cfg is not None
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 1, stride=1, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, stride=stride, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, planes, planes * self.expansion, 1, bias=False)
        self.add_module(self.norm3_name, norm3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.inplanes, self.planes, self.stride = inplanes, planes, stride

    norm1 = property(lambda self: getattr(self, self.norm1_name))
    norm2 = property(lambda self: getattr(self, self.norm2_name))
    norm3 = property(lambda self: getattr(self, self.norm3_name))

    def forward(self, x):
        identity = x.features
        out = self.conv1(x)
        out = replace_feature(out, self.relu(self.norm1(out.features)))
        out = self.conv2(out)
        out = replace_feature(out, self.relu(self.norm2(out.features)))
        out = self.conv3(out)
        out = replace_feature(out, self.norm3(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = replace_feature(out, out.features + identity)
        return replace_feature(out, self.relu(out.features))


class AdaptiveSparseBasicBlock(SparseBasicBlock):
    """SparseBasicBlock on ``planes`` channels with an adaptive front (sparse_block.py:146-213): when the channel count
    changes or the block strides, a regular sparse conv (kernel = stride with ``merge``, else 3 with padding 1) + norm +
    ReLU brings the input to ``planes`` channels and the strided sites first."""

    def __init__(self, inplanes, planes, stride=1, merge=True, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__(planes, planes, stride=1, downsample=downsample, conv_cfg=conv_cfg, norm_cfg=norm_cfg)
        is_stride = max(stride) > 1 if isinstance(stride, (tuple, list)) else stride > 1
        if inplanes != planes or is_stride:
            ndim = int(conv_cfg['type'][-2])
            assert ndim in (1, 2, 3, 4)
            ada_conv_cfg = dict(type=f'SparseConv{ndim}d', indice_key=conv_cfg['indice_key'] + '.adaptive')
            if merge:
                self.ada_conv = build_conv_layer(ada_conv_cfg, inplanes, planes, stride, stride=stride, padding=0)
            else:
                self.ada_conv = build_conv_layer(ada_conv_cfg, inplanes, planes, 3, stride=stride, padding=1)
            self.ada_norm = build_norm_layer(norm_cfg, planes)[1]
            self.ada_relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if hasattr(self, 'ada_conv'):
            x = self.ada_conv(x)
            x = replace_feature(x, self.ada_relu(self.ada_norm(x.features)))
        return super().forward(x)
