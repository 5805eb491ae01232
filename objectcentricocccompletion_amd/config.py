"""Config loading with the semantics of mmcv.Config.fromfile that matter for
configs/ococc/ococcnet.py: python-file configs, ``_base_`` inheritance with recursive dict
merge, and -- crucially -- every nested list/dict REBUILT on load the way mmcv's addict
ConfigDict does.  ococcnet.py writes ``rel_mlp_hidden_dims=[[16, 32]] * 6`` (six references to
one list) and SIRLayer appends to its argument (voxel_encoder.py:733); without the rebuild the
model silently has 66,927,378 instead of 66,553,173 parameters."""
import copy
import os


class ConfigDict(dict):
    """dict with attribute access (cfg.model.roi_head...), as mmcv's ConfigDict."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def rebuild(obj):
    """Deep copy that gives every nested list / tuple / dict its own identity."""
    if isinstance(obj, dict):
        return ConfigDict({k: rebuild(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [rebuild(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(rebuild(v) for v in obj)
    return copy.copy(obj)


def _merge(base, new):
    out = dict(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'} if isinstance(v, dict) else v
    return out


def _load_py(path):
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), scope)
    return {k: v for k, v in scope.items() if not k.startswith('__') and not callable(v)
            and not isinstance(v, type(os))}


def fromfile(path):
    """Load a python config file (with optional ``_base_`` list) into a ConfigDict."""
    path = os.path.abspath(path)
    cfg = _load_py(path)
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bpath = os.path.join(os.path.dirname(path), b)
        if os.path.exists(bpath):  # dataset / schedule bases may be absent next to a lone model config
            merged = _merge(merged, dict(fromfile(bpath)))
    return rebuild(_merge(merged, cfg))


def merge_from_dict(cfg, options):
    """--cfg-options style dotted overrides (tools/train.py:63-72)."""
    for key, value in options.items():
        d = cfg
        parts = key.split('.')
        for p in parts[:-1]:
            d = d.setdefault(p, ConfigDict())
        d[parts[-1]] = value
    return cfg
