"""Minimal registry / builder layer with the type strings and call conventions of the
mmcv registries the reference relies on (mmdet3d/models/builder.py:5-57, mmcv
CONV_LAYERS / NORM_LAYERS used by mmdet3d/ops/spconv/conv.py:207 and
mmdet3d/ops/norm.py:28).  mmcv itself is absent from the image, and the model code is
written against this shim so configs/ococc/ococcnet.py builds verbatim."""
import copy

from torch import nn


class Registry(object):

    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if not force and key in self._module_dict:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, default_args=None):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'{self.name}: cfg must be a dict with a "type" key, got {cfg!r}')
        args = copy.copy(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        typ = args.pop('type')
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        return cls(**args)


CONV_LAYERS = Registry('conv layer')
NORM_LAYERS = Registry('norm layer')
DETECTORS = Registry('detector')
HEADS = Registry('head')
BACKBONES = Registry('backbone')
ROI_EXTRACTORS = Registry('roi extractor')
VOXEL_ENCODERS = Registry('voxel encoder')
MIDDLE_ENCODERS = VOXEL_ENCODERS
MODELS = VOXEL_ENCODERS
LOSSES = Registry('loss')
BBOX_ASSIGNERS = Registry('bbox assigner')
BBOX_CODERS = Registry('bbox coder')
DATASETS = Registry('dataset')    # mmdet.datasets.DATASETS (waymo_tracklet_dataset.py:30)
PIPELINES = Registry('pipeline')  # mmdet.datasets.builder.PIPELINES (tracklet_pipelines.py:24)

CONV_LAYERS.register_module('Conv1d', module=nn.Conv1d)
CONV_LAYERS.register_module('Conv2d', module=nn.Conv2d)
CONV_LAYERS.register_module('Conv3d', module=nn.Conv3d)


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv.cnn.build_conv_layer: cfg = dict(type=..., **layer kwargs) or None (Conv2d)."""
    cfg = dict(type='Conv2d') if cfg is None else dict(cfg)
    typ = cfg.pop('type')
    cls = CONV_LAYERS.get(typ)
    if cls is None:
        raise KeyError(f'Unrecognized conv type {typ}')
    return cls(*args, **kwargs, **cfg)


def build_norm_layer(cfg, num_features, postfix=''):
    """mmcv.cnn.build_norm_layer -> (name, layer).  'LN' -> LayerNorm, 'BN1d' ->
    BatchNorm1d, 'naiveSyncBN1d' -> the synced variant of mmdet3d/ops/norm.py."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    cls = NORM_LAYERS.get(typ)
    if cls is None:
        raise KeyError(f'Unrecognized norm type {typ}')
    abbr = {'LN': 'ln', 'BN1d': 'bn', 'BN': 'bn', 'BN2d': 'bn', 'naiveSyncBN1d': 'bn'}.get(typ, 'norm')
    if 'eps' not in cfg and typ != 'LN':
        cfg.setdefault('eps', 1e-5)
    layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


NORM_LAYERS.register_module('BN', module=nn.BatchNorm2d)
NORM_LAYERS.register_module('BN1d', module=nn.BatchNorm1d)
NORM_LAYERS.register_module('BN2d', module=nn.BatchNorm2d)
NORM_LAYERS.register_module('BN3d', module=nn.BatchNorm3d)
