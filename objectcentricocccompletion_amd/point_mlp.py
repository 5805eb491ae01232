"""Fused per-point layers of the SIR encoders on csrc/point_mlp.hip:

    y = act(LayerNorm(x W^T)),   x = [ a * mul * colscale | b * bscale | v[inv] ],   optionally seg_max(y)

one launch where SIRLayer.forward (mmdet3d/models/voxel_encoders/voxel_encoder.py:764-832) runs torch.cat / products,
nn.Linear, LayerNorm, GELU, scatter_v2(max) and the gather-back one after the other.  f32 like the reference
(force_fp32); the backward launch recomputes the layer from its inputs."""
import os
import weakref

import torch

from . import _deferred
from . import _lib as L
from . import norm as _norm  # noqa: F401 (registers the 'ln' flusher of _deferred)
from .linear import TALL_ROWS, sliced_wgrad

ACT = {None: 0, 'none': 0, 'gelu': 1, 'relu': 2}

if os.environ.get('OCOCC_POINT_TILE'):   # 32 / 64: pin the rows per workgroup tile of the fwd / bwd launches (default: by input size)
    L.check(L.lib.ococc_point_mlp_force_tile(int(os.environ['OCOCC_POINT_TILE'])), 'point_mlp_force_tile')


_packed = {}   # (id of the parameter, transposed) -> (weak reference, version, storage pointer, fragments)


def pack_weight(w, owner=None, transposed=False):
    """nn.Linear weight [n, k] (or a transposed view) -> f32 MFMA fragment tensor.  With ``owner`` (the parameter the
    view was taken from) the fragments are cached until the parameter is written to (version counter), moves or dies --
    identity by weak reference: a new tensor at a recycled address is a different tensor."""
    n, k = w.shape
    key = None
    if owner is not None and not torch.cuda.is_current_stream_capturing():
        key = (id(owner), bool(transposed))
        hit = _packed.get(key)
        if hit is not None and hit[0]() is owner and hit[1] == owner._version and hit[2] == w.data_ptr():
            return hit[3]
    out = torch.empty(int(L.lib.ococc_point_mlp_fragment_floats(n, k)), dtype=torch.float32, device=w.device)
    L.check(L.lib.ococc_point_mlp_pack_f32(L.ptr(w), n, k, w.stride(0), w.stride(1), L.ptr(out), L.stream()), 'point_mlp_pack')
    if key is not None:
        if len(_packed) > 512:
            _packed.clear()
        _packed[key] = (weakref.ref(owner), owner._version, w.data_ptr(), out)
    return out


def prepack(weights, backward=True):
    """Pack the f32 weights of several layers (and, with ``backward``, their transposed views) in ONE launch, into the
    cache pack_weight reads: a SIR layer calls this with all its Linears before it runs them."""
    import ctypes
    if torch.cuda.is_current_stream_capturing():
        return
    todo = []
    for w in weights:
        if w.dtype != torch.float32 or not w.is_cuda:
            continue
        for tr in ((False, True) if backward else (False,)):
            key = (id(w), tr)
            hit = _packed.get(key)
            view = w.detach().t() if tr else w.detach()
            if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == view.data_ptr():
                continue
            todo.append((key, w, view))
    for lo in range(0, len(todo), 32):
        part = todo[lo:lo + 32]
        outs = [torch.empty(int(L.lib.ococc_point_mlp_fragment_floats(v.shape[0], v.shape[1])), dtype=torch.float32,
                            device=v.device) for _, _, v in part]
        c = len(part)
        vp, i32, i64 = ctypes.c_void_p * c, ctypes.c_int32 * c, ctypes.c_int64 * c
        L.check(L.lib.ococc_point_mlp_pack_multi_f32(
            c, vp(*[v.data_ptr() for _, _, v in part]), i32(*[v.shape[0] for _, _, v in part]),
            i32(*[v.shape[1] for _, _, v in part]), i64(*[v.stride(0) for _, _, v in part]),
            i64(*[v.stride(1) for _, _, v in part]), vp(*[o.data_ptr() for o in outs]), L.stream()), 'point_mlp_pack_multi')
        if len(_packed) > 512:
            _packed.clear()
        for (key, w, v), o in zip(part, outs):
            _packed[key] = (weakref.ref(w), w._version, v.data_ptr(), o)


class PackPlan(object):
    """prepack for a fixed set of parameters (the Linears of one SIR layer) with everything that does not change from
    step to step built once: the argument arrays of the pack launch and the fragment buffers themselves, rewritten in
    place when a parameter's version moves (stream order keeps earlier readers ahead of the rewrite; a graph that still
    needs the OLD values of a parameter written to in place is an error autograd reports on its own).  ~10 us of host
    time per call instead of ~80."""

    def __init__(self, weights, private=False):
        import ctypes
        self.private = private   # the owner reads self.outs itself: no need for the entries in pack_weight's cache to be ours
        self.weights = list(weights)
        self.views = []
        for w in self.weights:
            d = w.detach()
            self.views += [(w, False, d), (w, True, d.t())]
        self.outs = [torch.empty(int(L.lib.ococc_point_mlp_fragment_floats(v.shape[0], v.shape[1])), dtype=torch.float32,
                                 device=v.device) for _, _, v in self.views]
        self.ptrs = tuple(w.data_ptr() for w in self.weights)
        self.state = None   # (versions, with transposes) of the last pack
        self._args = {}
        for tr_too in (False, True):
            idx = [i for i, (_, tr, _) in enumerate(self.views) if tr_too or not tr]
            c = len(idx)
            vp, i32, i64 = ctypes.c_void_p * c, ctypes.c_int32 * c, ctypes.c_int64 * c
            vs = [self.views[i][2] for i in idx]
            self._args[tr_too] = (c, vp(*[v.data_ptr() for v in vs]), i32(*[v.shape[0] for v in vs]),
                                  i32(*[v.shape[1] for v in vs]), i64(*[v.stride(0) for v in vs]),
                                  i64(*[v.stride(1) for v in vs]), vp(*[self.outs[i].data_ptr() for i in idx]), idx)

    def valid_for(self, weights):
        return (len(weights) == len(self.weights) and all(a is b for a, b in zip(weights, self.weights))
                and tuple(w.data_ptr() for w in weights) == self.ptrs)

    def refresh(self, backward=True):
        if torch.cuda.is_current_stream_capturing():
            return
        versions = tuple(w._version for w in self.weights)
        if self.state is not None and self.state[0] == versions and (self.state[1] or not backward):
            # (still ours in the cache?  pack_weight may have cleared it)
            first = _packed.get((id(self.weights[0]), False))
            if self.private or (first is not None and first[3] is self.outs[0]):
                return
        c, src, n, k, s0, s1, dst, idx = self._args[bool(backward)]
        for lo in range(0, c, 32):
            if c <= 32:
                L.check(L.lib.ococc_point_mlp_pack_multi_f32(c, src, n, k, s0, s1, dst, L.stream()), 'point_mlp_pack_multi')
            else:   # (more than one launch: slices of the argument arrays)
                import ctypes
                m = min(32, c - lo)
                sl = lambda arr, ty: (ty * m)(*arr[lo:lo + m])
                L.check(L.lib.ococc_point_mlp_pack_multi_f32(
                    m, sl(src, ctypes.c_void_p), sl(n, ctypes.c_int32), sl(k, ctypes.c_int32), sl(s0, ctypes.c_int64),
                    sl(s1, ctypes.c_int64), sl(dst, ctypes.c_void_p), L.stream()), 'point_mlp_pack_multi')
        if len(_packed) > 512:
            _packed.clear()
        for i in idx:
            w, tr, v = self.views[i]
            _packed[(id(w), tr)] = (weakref.ref(w), w._version, v.data_ptr(), self.outs[i])
        self.state = (versions, bool(backward))


def refresh_plans(plans, backward=True):
    """PackPlan.refresh for several private plans at once: the stale ones share launches of 32 matrices (the twelve plans of
    a SIR stack -- 60 matrices -- are two launches per step instead of twelve)."""
    if torch.cuda.is_current_stream_capturing():
        return
    import ctypes
    stale = []
    for p in plans:
        versions = tuple(w._version for w in p.weights)
        if p.private and p.state is not None and p.state[0] == versions and (p.state[1] or not backward):
            continue
        stale.append((p, versions))
    if not stale:
        return
    cols = [[], [], [], [], [], []]   # src, n, k, stride 0, stride 1, dst
    for p, _ in stale:
        c, src, n, k, s0, s1, dst, _ = p._args[bool(backward)]
        for col, arr in zip(cols, (src, n, k, s0, s1, dst)):
            col.extend(arr[:c])
    total = len(cols[0])
    for lo in range(0, total, 32):
        m = min(32, total - lo)
        sl = lambda col, ty: (ty * m)(*col[lo:lo + m])
        L.check(L.lib.ococc_point_mlp_pack_multi_f32(
            m, sl(cols[0], ctypes.c_void_p), sl(cols[1], ctypes.c_int32), sl(cols[2], ctypes.c_int32), sl(cols[3], ctypes.c_int64),
            sl(cols[4], ctypes.c_int64), sl(cols[5], ctypes.c_void_p), L.stream()), 'point_mlp_pack_multi')
    for p, versions in stale:
        if not p.private:
            for i in p._args[bool(backward)][7]:
                w, tr, v = p.views[i]
                _packed[(id(w), tr)] = (weakref.ref(w), w._version, v.data_ptr(), p.outs[i])
        p.state = (versions, bool(backward))


def _f32(t):
    return None if t is None else t.detach().float().contiguous()


def layer_forward(a_, mul_, b_, v_, wf, n, g, be, colscale, inv, bscale, eps, act, want_max, num_segments):
    """The forward launch on prepared operands (f32, row-contiguous; ``wf`` the packed weight): (y, segment maxima)."""
    rows, ka = a_.shape
    kb = 0 if b_ is None else b_.shape[1]
    kv = 0 if v_ is None else v_.shape[1]
    dev = a_.device
    y = torch.empty((rows, n), dtype=torch.float32, device=dev)
    vmax = torch.empty((num_segments, n), dtype=torch.float32, device=dev) if want_max else None
    L.check(L.lib.ococc_point_mlp_fwd_f32(
        a_.data_ptr(), ka, a_.stride(0), L.ptr(mul_), 0 if mul_ is None else mul_.stride(0), L.ptr(colscale), L.ptr(b_), kb,
        0 if b_ is None else b_.stride(0), bscale, L.ptr(v_), kv, L.ptr(inv), rows, wf.data_ptr(), n, L.ptr(g),
        L.ptr(be), eps, ACT[act], y.data_ptr(), L.ptr(vmax), num_segments, L.stream()), 'point_mlp_fwd')
    return y, vmax


def layer_backward(a_, mul_, b_, v_, weight, wf, g, be, colscale, inv, y, vmax, bscale, eps, act, G, dy, dvmax,
                   need_a, need_mul, need_b, need_v, need_w, weight_param=None):
    """The backward launch (+ the weight gradient): (da, dmul, db, dv, dw, ln partials [tiles, 2, n], tiles).
    ``dy`` / ``dvmax``: f32 contiguous or None (dvmax); ``weight_param``: the parameter itself when the caller's saved
    tensor is not it (its gradient may then be queued for the end of the pass: dw is None)."""
    rows, ka = a_.shape
    kb = 0 if b_ is None else b_.shape[1]
    kv = 0 if v_ is None else v_.shape[1]
    n, k = weight.shape
    dev = a_.device
    arg = None
    if vmax is not None and dvmax is not None:
        arg = torch.empty((G, n), dtype=torch.int32, device=dev)
        L.check(L.lib.ococc_point_mlp_segment_argmax(y.data_ptr(), vmax.data_ptr(), inv.data_ptr(), rows, n, G, arg.data_ptr(),
                                                     L.stream()), 'segment_argmax')
    else:
        dvmax = None
    wtf = pack_weight(weight.detach().float().t(), weight if weight.dtype == torch.float32 else None, transposed=True)
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    dz = new(rows, n)
    xcat = new(rows, k) if need_w else None
    da = new(rows, ka) if need_a else None
    dmul = new(rows, ka) if (mul_ is not None and need_mul) else None
    db = new(rows, kb) if (b_ is not None and need_b) else None
    dv = torch.zeros((G, kv), dtype=torch.float32, device=dev) if (v_ is not None and need_v) else None
    tiles = int(L.lib.ococc_point_mlp_tiles(rows))
    lnp = new(tiles, 2, n) if g is not None else None
    L.check(L.lib.ococc_point_mlp_bwd_f32(
        a_.data_ptr(), ka, a_.stride(0), L.ptr(mul_), 0 if mul_ is None else mul_.stride(0), L.ptr(colscale), L.ptr(b_), kb,
        0 if b_ is None else b_.stride(0), bscale, L.ptr(v_), kv, L.ptr(inv), rows, wf.data_ptr(), wtf.data_ptr(), n, L.ptr(g),
        L.ptr(be), eps, ACT[act], dy.data_ptr(), L.ptr(dvmax), L.ptr(arg), dz.data_ptr(), L.ptr(xcat), L.ptr(da), L.ptr(dmul),
        L.ptr(db), L.ptr(dv), L.ptr(lnp), L.stream()), 'point_mlp_bwd')
    dw = None
    if need_w:
        dw = weight_grad(weight_param if weight_param is not None else weight, dz, xcat)
    return da, dmul, db, dv, dw, lnp, tiles


WGRAD_KERNEL = os.environ.get('OCOCC_POINT_WGRAD', '1') == '1'


def weight_grad(weight, dz, xcat):
    """dW = dz^T x_cat [n, k]: per-row-slice products from ococc_point_mlp_wgrad_f32, summed at the end of the backward
    pass together with the LayerNorm parameter sums (None is returned then: the sum reaches weight.grad there) or right
    away; OCOCC_POINT_WGRAD=0: the library GEMMs."""
    rows, n = dz.shape
    k = xcat.shape[1]
    if not WGRAD_KERNEL or rows == 0 or (n * k) % 2:
        return (sliced_wgrad(dz, xcat) if rows >= 4096 else dz.t() @ xcat).to(weight.dtype)
    slices = int(L.lib.ococc_point_mlp_wgrad_slices(rows))
    partial = torch.empty((slices, n, k), dtype=torch.float32, device=dz.device)
    L.check(L.lib.ococc_point_mlp_wgrad_f32(dz.data_ptr(), xcat.data_ptr(), rows, n, k, partial.data_ptr(), L.stream()),
            'point_mlp_wgrad')
    if weight.dtype == torch.float32 and weight.shape == (n, k) and _deferred.deferrable(weight):
        out = torch.empty((2, n * k // 2), dtype=torch.float32, device=dz.device)
        if _deferred.defer('ln', (partial, slices, n * k // 2, out), [(weight, out.view(n, k))]):
            return None
    return (partial.sum(0) if slices > 1 else partial[0]).to(weight.dtype)


def ln_param_grads(ln_w, ln_b, lnp, tiles, n, need):
    """(d gamma, d beta) of one layer from its per-tile partial rows [tiles][d gamma | d beta] -- the layout of the
    LayerNorm kernels' partials, so the column sums can ride on the pass's end-of-backward launch (_deferred,
    norm._flush_param_reduce) instead of one reduction launch per layer; (None, None) then."""
    if (tiles > 0 and need and ln_w is not ln_b and _deferred.deferrable(ln_w, ln_b)):
        dgb = torch.empty((2, n), dtype=torch.float32, device=lnp.device)
        if _deferred.defer('ln', (lnp, tiles, n, dgb), [(ln_w, dgb[0]), (ln_b, dgb[1])]):
            return None, None
    sums = lnp.sum(0)
    return sums[0], sums[1]


class _PointLayer(torch.autograd.Function):

    @staticmethod
    def forward(ctx, a, mul, b, v, weight, ln_w, ln_b, colscale, inv, bscale, eps, act, want_max, num_segments):
        L.require_device(a, weight)
        a_, mul_, b_, v_ = _f32(a), _f32(mul), _f32(b), _f32(v)
        n = weight.shape[0]
        assert weight.shape[1] == a_.shape[1] + (0 if b_ is None else b_.shape[1]) + (0 if v_ is None else v_.shape[1]), \
            (weight.shape, a_.shape)
        own = weight if weight.dtype == torch.float32 else None
        wf = pack_weight(weight.detach().float(), own)
        g, be = _f32(ln_w), _f32(ln_b)
        y, vmax = layer_forward(a_, mul_, b_, v_, wf, n, g, be, colscale, inv, float(bscale), float(eps), act, want_max,
                                int(num_segments))
        ctx.save_for_backward(a_, mul_, b_, v_, weight, g, be, colscale, inv, y, vmax, wf)
        ctx.ln_params = (ln_w, ln_b)   # the parameters themselves: their gradient sums may join the end-of-backward launch
        ctx.weight_param = weight
        ctx.misc = (float(bscale), float(eps), act, int(num_segments))
        ctx.in_dtypes = tuple(None if t is None else t.dtype for t in (a, mul, b, v))
        return y, vmax

    @staticmethod
    def backward(ctx, dy, dvmax):
        a_, mul_, b_, v_, weight, g, be, colscale, inv, y, vmax, wf = ctx.saved_tensors
        bscale, eps, act, G = ctx.misc
        need = ctx.needs_input_grad
        da, dmul, db, dv, dw, lnp, tiles = layer_backward(
            a_, mul_, b_, v_, weight, wf, g, be, colscale, inv, y, vmax, bscale, eps, act, G, _f32(dy), _f32(dvmax),
            need[0], need[1], need[2], need[3], need[4], ctx.weight_param)
        dg = dbeta = None
        if g is not None:
            ln_w, ln_b = ctx.ln_params
            dg, dbeta = ln_param_grads(ln_w, ln_b, lnp, tiles, weight.shape[0], need[5] and need[6])
        cast = lambda t, dt: None if t is None else t.to(dt)
        dts = ctx.in_dtypes
        return (cast(da, dts[0]), cast(dmul, dts[1]), cast(db, dts[2]), cast(dv, dts[3]), dw, dg, dbeta, None, None, None,
                None, None, None, None)


def point_layer(a, weight, ln_weight=None, ln_bias=None, eps=1e-5, act='gelu', mul=None, colscale=None, b=None, bscale=1.0,
                v=None, inv=None, num_segments=0, seg_max=False):
    """y [rows, n] (and the segment maxima [num_segments, n] when ``seg_max``).  ``inv``: int32 segment of every row,
    non-decreasing; needed for ``v`` (rows of a per-segment tensor appended to the input) and for ``seg_max``."""
    if (v is not None or seg_max) and inv is None:
        raise ValueError('inv is needed to gather segment rows / reduce over segments')
    if inv is not None and inv.dtype != torch.int32:
        inv = inv.to(torch.int32)
    y, vmax = _PointLayer.apply(a, mul, b, v, weight, ln_weight, ln_bias, colscale, inv, bscale, eps, act, bool(seg_max),
                                int(num_segments))
    return (y, vmax) if seg_max else y
