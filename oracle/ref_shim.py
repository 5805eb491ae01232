"""Import the reference's hot-path Python files IN THE BUILD CONTAINER ONLY.

mmcv / mmdet / mmseg / torch_scatter / TorchEx / spconv are not installed here, so the
reference package cannot be imported as a whole (mmdet3d/__init__.py:1-3).  This shim
follows SURVEY.md Appendix A: it pre-registers empty package shells whose __path__ points
into /root/reference (so no heavyweight __init__.py runs), fabricates stub modules for the
absent third-party roots, and provides real stand-ins for the ~10 symbols the hot path
touches (identity decorators, nn.Module as BaseModule, LayerNorm factory, tiny registries,
torch_scatter emulated with Tensor.scatter_reduce).  The arithmetic under test stays in the
reference's own files.

Used only by oracle/gen_golden_*.py and tests that are skipped when /root/reference is
absent.  Nothing here ships to the product and no reference source is copied.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch
from torch import nn

REF = os.environ.get('OCOCC_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'mmdet3d'))


class _Registry(object):
    def __init__(self, name):
        self.name = name
        self.d = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.d[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                cfg.setdefault(k, v)
        return self.d[cfg.pop('type')](**cfg)

    def get(self, k):
        return self.d.get(k)


class _AutoStub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        v = types.SimpleNamespace()
        setattr(self, name, v)
        return v


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ('mmcv', 'mmdet', 'mmseg', 'ipdb', 'torch_scatter', 'ingroup_indices',
             'dynamic_point_pool_ext', 'spconv', 'numba', 'boxes_overlap_1to1')

    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _AutoStub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _identity_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def deco(fn):
        return fn
    return deco


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


def _build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    cfg.pop('requires_grad', None)
    if typ == 'LN':
        return 'ln', nn.LayerNorm(num_features, **cfg)
    if typ in ('BN1d', 'naiveSyncBN1d'):
        return 'bn', nn.BatchNorm1d(num_features, **cfg)
    raise KeyError(typ)


def _scatter_max(src, index, dim=0, out=None, dim_size=None):
    n = int(index.max()) + 1 if dim_size is None else dim_size
    res = src.new_full((n,) + tuple(src.shape[1:]), float('-inf'))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    res = res.scatter_reduce(0, idx, src, 'amax', include_self=True)
    return res, torch.zeros_like(res, dtype=torch.long)


def _scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    n = int(index.max()) + 1 if dim_size is None else dim_size
    res = src.new_zeros((n,) + tuple(src.shape[1:]))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return res.scatter_reduce(0, idx, src, {'sum': 'sum', 'mean': 'mean', 'max': 'amax'}[reduce],
                              include_self=False)


_installed = False
REG = {}


def install():
    """Idempotent: put the shells / stubs into sys.modules."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f'{REF} is not present: the reference can only be imported in the build container')
    sys.meta_path.insert(0, _StubFinder())
    shells = ['mmdet3d', 'mmdet3d.ops', 'mmdet3d.ops.sst', 'mmdet3d.ops.occ', 'mmdet3d.ops.iou3d',
              'mmdet3d.ops.roiaware_pool3d', 'mmdet3d.models', 'mmdet3d.models.occ',
              'mmdet3d.models.backbones', 'mmdet3d.models.voxel_encoders', 'mmdet3d.models.roi_heads',
              'mmdet3d.models.roi_heads.bbox_heads', 'mmdet3d.models.sst', 'mmdet3d.models.middle_encoders',
              'mmdet3d.core', 'mmdet3d.core.bbox', 'mmdet3d.core.bbox.structures', 'mmdet3d.core.bbox.coders',
              'mmdet3d.core.points', 'mmdet3d.core.post_processing']
    for name in shells:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, *name.split('.'))]
        sys.modules[name] = m
    for name in shells:  # make `import a.b.c` / `from a.b import c` resolve through parents
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            setattr(sys.modules[parent], child, sys.modules[name])

    import mmcv.runner  # noqa  (fabricated)
    import mmcv.cnn  # noqa
    import mmdet.models  # noqa
    import mmdet.core  # noqa
    import torch_scatter  # noqa
    sys.modules['mmcv.runner'].force_fp32 = _identity_decorator
    sys.modules['mmcv.runner'].auto_fp16 = _identity_decorator
    sys.modules['mmcv.runner'].BaseModule = _BaseModule
    sys.modules['mmcv.cnn'].build_norm_layer = _build_norm_layer
    for r in ('HEADS', 'BACKBONES', 'DETECTORS', 'ROI_EXTRACTORS', 'LOSSES', 'NECKS'):
        REG[r] = _Registry(r)
        setattr(sys.modules['mmdet.models'], r, REG[r])
    REG['VOXEL_ENCODERS'] = _Registry('VOXEL_ENCODERS')
    sys.modules['mmdet.core'].reduce_mean = lambda t: t
    sys.modules['mmdet.core'].multi_apply = lambda f, *a, **k: tuple(map(list, zip(*map(lambda *x: f(*x, **k), *a))))

    import mmdet.core.bbox  # noqa (fabricated)
    import mmdet.core.bbox.builder  # noqa
    sys.modules['mmdet.core.bbox'].BaseBBoxCoder = object
    REG['BBOX_CODERS'] = _Registry('BBOX_CODERS')
    sys.modules['mmdet.core.bbox.builder'].BBOX_CODERS = REG['BBOX_CODERS']

    def _build_coder(cfg):
        importlib.import_module('mmdet3d.core.bbox.coders.delta_xyzwhlr_bbox_coder')  # the reference's own coder
        return REG['BBOX_CODERS'].build(cfg)
    sys.modules['mmdet.core'].build_bbox_coder = _build_coder
    sys.modules['torch_scatter'].scatter_max = _scatter_max
    sys.modules['torch_scatter'].scatter = _scatter

    ops = sys.modules['mmdet3d.ops']
    ops.spconv = types.SimpleNamespace()
    ops.DynamicScatter = lambda *a, **k: None
    ops.make_sparse_convmodule = None  # imported by voxel_encoder.py:12, unused on the ococc path
    # mmdet3d.models.builder stand-in
    b = types.ModuleType('mmdet3d.models.builder')
    b.VOXEL_ENCODERS = REG['VOXEL_ENCODERS']
    b.build_voxel_encoder = REG['VOXEL_ENCODERS'].build
    b.build_backbone = REG['BACKBONES'].build
    b.build_head = REG['HEADS'].build
    b.build_loss = lambda cfg: None
    b.build_roi_extractor = REG['ROI_EXTRACTORS'].build
    b.build_fusion_layer = lambda cfg: None
    sys.modules['mmdet3d.models.builder'] = b
    sys.modules['mmdet3d.models'].builder = b
    n = types.ModuleType('mmdet3d.ops.norm')
    n.AllReduce = None
    sys.modules['mmdet3d.ops.norm'] = n
    iu = types.ModuleType('mmdet3d.ops.iou3d.iou3d_utils')
    iu.nms_gpu = iu.nms_normal_gpu = None
    sys.modules['mmdet3d.ops.iou3d.iou3d_utils'] = iu

    sst_ops = importlib.import_module('mmdet3d.ops.sst.sst_ops')
    for name in ('scatter_v2', 'build_mlp', 'get_activation_layer', 'get_activation', 'get_inner_win_inds',
                 'get_inner_win_inds_deprecated', 'flat2window', 'window2flat', 'flat2window_v2',
                 'window2flat_v2', 'get_flat2win_inds', 'get_flat2win_inds_v2', 'get_window_coors',
                 'make_continuous_inds'):
        if hasattr(sst_ops, name):
            setattr(ops, name, getattr(sst_ops, name))
    occ_ops = importlib.import_module('mmdet3d.ops.occ.occ_ops')
    sys.modules['mmdet3d.ops.occ'].occ_ops = occ_ops
    su = importlib.import_module('mmdet3d.core.bbox.structures.utils')
    st = sys.modules['mmdet3d.core.bbox.structures']
    st.rotation_3d_in_axis = su.rotation_3d_in_axis
    st.xywhr2xyxyr = getattr(su, 'xywhr2xyxyr', None)
    st.LiDARInstance3DBoxes = object
    _installed = True


def load(dotted):
    install()
    return importlib.import_module(dotted)


def load_ococc_classes():
    """SIRLayer, SIR, OccDecoder/PosEncode, layers, OccAutoEncoder, OccBBoxHead (+ helpers)."""
    install()
    ve = load('mmdet3d.models.voxel_encoders.voxel_encoder')
    sir = load('mmdet3d.models.backbones.sir')
    occ_base = load('mmdet3d.models.occ.occ_base')
    layers = load('mmdet3d.models.occ.layers')
    fsd = load('mmdet3d.models.roi_heads.bbox_heads.fsd_bbox_head')
    sys.modules['mmdet3d.models.roi_heads.bbox_heads'].FullySparseBboxHead = fsd.FullySparseBboxHead
    ae = load('mmdet3d.models.roi_heads.bbox_heads.occ_ae_head')
    head = load('mmdet3d.models.roi_heads.bbox_heads.ococc_bbox_head')
    return dict(voxel_encoder=ve, sir=sir, occ_base=occ_base, layers=layers, fsd=fsd, ae=ae, head=head,
                sst_ops=sys.modules['mmdet3d.ops.sst.sst_ops'], occ_ops=sys.modules['mmdet3d.ops.occ.occ_ops'],
                utils=sys.modules['mmdet3d.core.bbox.structures.utils'])
