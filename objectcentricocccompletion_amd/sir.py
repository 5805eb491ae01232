"""SIR point encoders -- host mirror of SIRLayer (mmdet3d/models/voxel_encoders/
voxel_encoder.py:686-832), DynamicVFELayerV2 (voxel_encoders/utils.py:147-189) and SIR
(mmdet3d/models/backbones/sir.py:16-88).  Same constructor arguments, parameter names
(rel_mlp.<i>.0.weight, vfe_layers.<j>.{linear,norm}.*, block_list.<i>...) and outputs.

Device work: every Linear -> LayerNorm -> activation of a layer, together with the products / concatenations /
gather-backs that build its input and the segment maximum behind it, is one launch of ococc_point_mlp_*_f32
(csrc/point_mlp.hip, f32 MFMA); options the kernel does not cover (batch norm, mean pooling, distance decoration)
run the same chain from separate operators.
"""
import ctypes
import os
import weakref

import torch
from torch import nn

from ._lib import const_tensor
from .linear import Linear
from . import _deferred
from . import _lib as L
from .point_mlp import PackPlan, layer_backward, layer_forward, ln_param_grads, pack_weight, point_layer, prepack, refresh_plans
from .registry import BACKBONES, VOXEL_ENCODERS, build_norm_layer
from .sst.sst_ops import build_mlp, fuse_norm_act, get_activation_layer, unique_with_inverse
from .voxel.scatter_points import gather_rows, segment_reduce

POINT_LAYER_KERNEL = True   # False: every Linear / LayerNorm / segment reduction of a SIRLayer as its own operator
# The fused layer is one workgroup per 64 points with all phases in sequence: it removes ~15 launches per layer (the
# config's own batch of 4 tracklets, ~8 k points, is launch bound) but at 1e5 points the separate library GEMM /
# LayerNorm / segment kernels, each tuned for throughput, are faster (measured on MI355X, whole ococcnet step: 25.4 / 35.8 / 51.1 ms fused against
# 33.5 / 41.9 / 53.6 ms at 4 / 16 / 32 tracklets = 8 k / 33 k / 65 k points, but 98 against 79 ms at 64 tracklets).  Above this many points the per-operator chain runs.
POINT_LAYER_MAX_ROWS = int(os.environ.get('OCOCC_POINT_LAYER_MAX_ROWS', 80000))
# One autograd node per SIRLayer (the launches are the same: at 4 tracklets the step is bound by the host, and a layer as
# five Functions + the concatenations between them costs ~0.4 ms of interpreter / autograd time per direction).
WHOLE_LAYER_NODE = os.environ.get('OCOCC_SIR_WHOLE_LAYER', '1') == '1'


_PACK_PLANS = weakref.WeakKeyDictionary()   # SIRLayer -> PackPlan of its Linears (not on the module: it holds ctypes arrays)
_NATIVE_PLANS = weakref.WeakKeyDictionary()   # SIRLayer -> _NativePlan


def _f32c(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _f32_rows(t):
    """t [rows, cols] as f32 with unit column stride and a row stride >= cols (a copy only if it is not that already)"""
    t = t.detach()
    if t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return t.float().contiguous()


# ... and one library call per direction for the whole layer (csrc/sir_layer.hip): ONE launch per direction + the
# weight-gradient launch while every row tile has a workgroup of its own (csrc/sir_fused_impl.hpp: <= 28 k points on
# MI355X; whole step at 4 tracklets 13.9 -> 13.2 ms), the launches the node below issues from Python otherwise.
NATIVE_LAYER = os.environ.get('OCOCC_SIR_NATIVE_LAYER', '1') == '1'


def check_barriers():
    """Raise if a grid barrier of a one-launch SIRLayer gave up since the last check (csrc/sir_fused.hip: not every
    workgroup of the persistent grid was resident -- two processes or streams with such grids on one device).  A host
    load of a word the device writes: no synchronisation, no copy.  The library makes the same test in front of every
    one-launch layer; callers add it where they have waited for the device anyway (the pooled-point read-back of every
    step: point_pool.py; tools/train.py per iteration and before a checkpoint; bench.py after its timed loop), so that a
    stranded barrier in the LAST layer of a pass is reported too."""
    L.check(L.lib.ococc_sir_layer_fused_check(), 'sir_layer barrier check')


class _SirLayerDesc(ctypes.Structure):   # ococc_sir_layer of include/ococc_hip.h
    _fields_ = [('n_rel', ctypes.c_int32), ('n_vfe', ctypes.c_int32), ('feat_cols', ctypes.c_int32),
                ('cluster_cols', ctypes.c_int32), ('with_cluster_center', ctypes.c_int32), ('shortcut', ctypes.c_int32),
                ('bscale', ctypes.c_float), ('inference', ctypes.c_int32), ('rel_colscale', ctypes.c_void_p),
                ('colscale', ctypes.c_void_p), ('n', ctypes.c_int32 * 8), ('act', ctypes.c_int32 * 8),
                ('eps', ctypes.c_float * 8), ('w_frag', ctypes.c_void_p * 8), ('wt_frag', ctypes.c_void_p * 8),
                ('ln_weight', ctypes.c_void_p * 8), ('ln_bias', ctypes.c_void_p * 8), ('gate', ctypes.c_void_p),
                ('dgate', ctypes.c_void_p)]


class _Ptr(object):
    """a device address inside a slab (which it keeps alive) -- what the end-of-pass reduction needs of a tensor"""
    __slots__ = ('base', 'p')

    def __init__(self, base, p):
        self.base, self.p = base, p

    def data_ptr(self):
        return self.p


class _NativePlan(object):
    """Per SIRLayer: the descriptor the library calls read (rebuilt when a parameter moves), the weight-fragment plan
    and the sizes that follow from the block shapes."""

    def __init__(self, blocks, n_rel, feat_cols, cluster_cols, with_cc, bscale, rel_cs, col):
        self.params = []
        for lin, norm, _ in blocks:
            self.params += [lin.weight, norm.weight, norm.bias]
        self.weights = [lin.weight for lin, _, _ in blocks]
        self.pack = PackPlan(self.weights, private=True)
        self.consts = (rel_cs, col)
        d = _SirLayerDesc()
        d.n_rel, d.n_vfe = n_rel, len(blocks) - n_rel
        d.feat_cols, d.cluster_cols, d.with_cluster_center, d.shortcut = feat_cols, cluster_cols, int(with_cc), 0
        d.bscale = bscale
        d.rel_colscale = None if rel_cs is None else rel_cs.data_ptr()
        d.colscale = None if col is None else col.data_ptr()
        for b, (lin, norm, act) in enumerate(blocks):
            d.n[b] = lin.out_features
            d.act[b] = {'none': 0, 'gelu': 1, 'relu': 2}[act]
            d.eps[b] = float(norm.eps)
            d.w_frag[b] = self.pack.outs[2 * b].data_ptr()
            d.wt_frag[b] = self.pack.outs[2 * b + 1].data_ptr()
            d.ln_weight[b] = norm.weight.data_ptr()
            d.ln_bias[b] = norm.bias.data_ptr()
        self.desc = d
        self.ref = ctypes.byref(d)
        self.ptrs = tuple(p.data_ptr() for p in self.params)
        self.nl = len(blocks)
        self.ns = [lin.out_features for lin, _, _ in blocks]
        self.ks = [lin.in_features for lin, _, _ in blocks]
        self.n_last = self.ns[-1]
        self.sum_n = sum(self.ns[n_rel:])
        # gradient buffers of one backward: per block [2, n] (LayerNorm) then [n k] (weight), one allocation
        self.grad_sizes = []
        for n, k in zip(self.ns, self.ks):
            self.grad_sizes += [n * k, n, n]       # the order of self.params
        n_vfe = len(blocks) - n_rel
        expect = [cluster_cols if j == 0 else self.ns[j - 1] for j in range(n_rel)]
        expect += [feat_cols + (cluster_cols if with_cc else 0)] + [2 * self.ns[n_rel + i - 1] for i in range(1, n_vfe)]
        self.ok = (expect == self.ks and all((n * k) % 2 == 0 for n, k in zip(self.ns, self.ks))
                   and all(p.dtype == torch.float32 and p.is_contiguous() for p in self.params)
                   and (n_rel == 0 or self.ns[n_rel - 1] == feat_cols))
        self._layouts = {}

    def valid_for(self, params):
        return len(params) == len(self.params) and all(a is b for a, b in zip(params, self.params)) \
            and tuple(p.data_ptr() for p in params) == self.ptrs

    def bwd_layout(self, rows, groups):
        key = (rows, groups)
        hit = self._layouts.get(key)
        if hit is None:
            ln_off, w_off = (ctypes.c_int64 * 8)(), (ctypes.c_int64 * 8)()
            tiles, slices, total = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int64()
            L.check(L.lib.ococc_sir_layer_bwd_layout(self.ref, rows, groups, ln_off, w_off, ctypes.byref(tiles),
                                                     ctypes.byref(slices), ctypes.byref(total)), 'sir_layer_bwd_layout')
            if len(self._layouts) > 64:
                self._layouts.clear()
            hit = self._layouts[key] = (list(ln_off)[:self.nl], list(w_off)[:self.nl], tiles.value, slices.value, total.value)
        return hit


class _SirLayerNative(torch.autograd.Function):
    """_SirLayerFn with the launch sequences inside the library (ococc_sir_layer_fwd_f32 / _bwd_f32)."""

    @staticmethod
    def forward(ctx, plan, shortcut, features, f_cluster, inv, G, gate, *params):
        # ``gate``: None, or (a plan without rel blocks) the layer's rel_mlp output computed elsewhere (rel_gates below)
        feats, fc = _f32c(features), _f32c(f_cluster)
        gate = None if gate is None else _f32c(gate)
        rows, dev = feats.shape[0], feats.device
        d = plan.desc
        d.shortcut = int(shortcut)
        d.inference = 0 if torch.is_grad_enabled() or any(ctx.needs_input_grad) else 1
        d.gate, d.dgate = (None if gate is None else gate.data_ptr()), None
        slab = torch.empty((int(L.lib.ococc_sir_layer_fwd_floats(plan.ref, rows, G)),), dtype=torch.float32, device=dev)
        y = torch.empty((rows, plan.n_last), dtype=torch.float32, device=dev)
        groups = torch.empty((G, plan.sum_n), dtype=torch.float32, device=dev)
        L.check(L.lib.ococc_sir_layer_fwd_f32(plan.ref, feats.data_ptr(), fc.data_ptr(), inv.data_ptr(), rows, G,
                                              slab.data_ptr(), y.data_ptr(), groups.data_ptr(), L.stream()), 'sir_layer_fwd')
        ctx.plan, ctx.shortcut, ctx.G = plan, bool(shortcut), G
        ctx.set_materialize_grads(False)
        ctx.has_gate = gate is not None
        ctx.save_for_backward(feats, fc, inv, slab, y, *(() if gate is None else (gate,)), *params)   # (the parameters: for autograd's version check)
        return y, groups

    @staticmethod
    def backward(ctx, dy, dM):
        plan, G = ctx.plan, ctx.G
        t = ctx.saved_tensors
        feats, fc, inv, fslab, y = t[:5]
        gate = t[5] if ctx.has_gate else None
        rows, dev = feats.shape[0], feats.device
        need = ctx.needs_input_grad
        d = plan.desc
        d.shortcut = int(ctx.shortcut)
        # (the descriptor is shared by every call on this plan: a no_grad forward of the same layer between this node's
        # forward and backward -- validation inside a step, a teacher pass -- leaves inference = 1 behind; THIS node's
        # forward recorded its arg-max rows, or there would be no backward)
        d.inference = 0
        dgate = None if gate is None else torch.empty_like(gate)
        d.gate, d.dgate = (None if gate is None else gate.data_ptr()), (None if dgate is None else dgate.data_ptr())
        ln_off, w_off, tiles, slices, total = plan.bwd_layout(rows, G)
        slab = torch.empty((total,), dtype=torch.float32, device=dev)
        dfeat = torch.empty_like(feats) if need[2] else None
        # (gradients that are column slices of a wider tensor -- the concatenations around the layer -- are read in place)
        dy = None if dy is None else _f32_rows(dy)
        dM = None if dM is None else _f32_rows(dM)
        L.check(L.lib.ococc_sir_layer_bwd_f32(plan.ref, feats.data_ptr(), fc.data_ptr(), inv.data_ptr(), rows, G,
                                              fslab.data_ptr(), y.data_ptr(), L.ptr(dy), dy.stride(0) if dy is not None else 0,
                                              L.ptr(dM), dM.stride(0) if dM is not None else 0, slab.data_ptr(),
                                              L.ptr(dfeat), L.stream()), 'sir_layer_bwd')
        if rows == 0:
            return (None, None, dfeat, None, None, None, dgate, *[torch.zeros_like(p) for p in plan.params])
        grads = _finish_param_grads(plan, slab, ln_off, w_off, tiles, slices, all(need[7:]))
        return (None, None, dfeat, None, None, None, dgate, *grads)


def _finish_param_grads(plan, slab, ln_off, w_off, tiles, slices, all_needed):
    """The parameter gradients of a plan's blocks from the partial sums a backward call left in ``slab`` (per block the
    weight-gradient slices [slices][n][k] and the LayerNorm partial rows [tiles][2][n]): queued for the ONE reduction launch
    at the end of the backward pass when the parameters allow it (then autograd gets None for them: the queue owns their
    .grad), summed right away otherwise.  In the order of plan.params: (weight, ln weight, ln bias) per block."""
    grads = [None] * len(plan.params)
    base, dev = slab.data_ptr(), slab.device
    if all_needed and _deferred.deferrable(*plan.params):
        out = torch.empty((sum(plan.grad_sizes),), dtype=torch.float32, device=dev)
        views = out.split(plan.grad_sizes)
        jobs, o = [], out.data_ptr()
        for b in range(plan.nl):
            n, k = plan.ns[b], plan.ks[b]
            half = n * k // 2
            jobs.append((_Ptr(slab, base + 4 * w_off[b]), slices, half, (_Ptr(out, o), _Ptr(out, o + 4 * half))))
            o += 4 * n * k
            jobs.append((_Ptr(slab, base + 4 * ln_off[b]), tiles, n, (_Ptr(out, o), _Ptr(out, o + 4 * n))))
            o += 8 * n
        if _deferred.defer_many('ln', jobs, list(zip(plan.params, views))):
            return grads
    for b in range(plan.nl):   # the sums right away, handed back through the engine
        n, k = plan.ns[b], plan.ks[b]
        grads[3 * b] = slab[w_off[b]: w_off[b] + slices * n * k].view(slices, n, k).sum(0)
        lnp = slab[ln_off[b]: ln_off[b] + tiles * 2 * n].view(tiles, 2, n).sum(0)
        grads[3 * b + 1], grads[3 * b + 2] = lnp[0], lnp[1]
    return grads


# ---- the rel_mlp chains of several layers in one launch per direction (csrc/sir_rel_chains.hip) ----------------------
BATCH_REL_CHAINS = os.environ.get('OCOCC_SIR_BATCH_REL', '1') == '1'
# (no row threshold of their own: three blocks of <= 32 input channels are skinny products the library runs at a few percent of
# the memory rate -- 52 + 52 + 65 us at 131 k rows, plus three LayerNorm launches, against 33 + 34 + 70 us for these kernels --
# so above POINT_LAYER_MAX_ROWS the gates still come from here and the vfe blocks run operator by operator)
REL_CHAINS_MAX_ROWS = int(os.environ.get('OCOCC_REL_CHAINS_MAX_ROWS', 4000000))
_REL_PLANS = weakref.WeakKeyDictionary()      # SIRLayer -> _RelPlan
_VFE_PLANS = weakref.WeakKeyDictionary()      # SIRLayer -> _NativePlan of its vfe blocks alone (the gate comes from outside)


class _RelChainDesc(ctypes.Structure):   # ococc_sir_rel_chain of include/ococc_hip.h
    _fields_ = [('n_blocks', ctypes.c_int32), ('cluster_cols', ctypes.c_int32), ('rel_colscale', ctypes.c_void_p),
                ('n', ctypes.c_int32 * 4), ('act', ctypes.c_int32 * 4), ('eps', ctypes.c_float * 4),
                ('w_frag', ctypes.c_void_p * 4), ('wt_frag', ctypes.c_void_p * 4), ('ln_weight', ctypes.c_void_p * 4),
                ('ln_bias', ctypes.c_void_p * 4)]


class _RelPlan(object):
    """Per SIRLayer: the chain descriptor of its rel_mlp (rebuilt when a parameter moves) and its weight-fragment plan."""

    def __init__(self, blocks, cluster_cols, rel_cs):
        self.params = []
        for lin, norm, _ in blocks:
            self.params += [lin.weight, norm.weight, norm.bias]
        self.pack = PackPlan([lin.weight for lin, _, _ in blocks], private=True)
        self.rel_cs = rel_cs
        d = _RelChainDesc()
        d.n_blocks, d.cluster_cols = len(blocks), cluster_cols
        d.rel_colscale = None if rel_cs is None else rel_cs.data_ptr()
        for b, (lin, norm, act) in enumerate(blocks):
            d.n[b] = lin.out_features
            d.act[b] = {'none': 0, 'gelu': 1, 'relu': 2}[act]
            d.eps[b] = float(norm.eps)
            d.w_frag[b] = self.pack.outs[2 * b].data_ptr()
            d.wt_frag[b] = self.pack.outs[2 * b + 1].data_ptr()
            d.ln_weight[b] = norm.weight.data_ptr()
            d.ln_bias[b] = norm.bias.data_ptr()
        self.desc = d
        self.ptrs = tuple(p.data_ptr() for p in self.params)
        self.nl = len(blocks)
        self.ns = [lin.out_features for lin, _, _ in blocks]
        self.ks = [lin.in_features for lin, _, _ in blocks]
        self.grad_sizes = []
        for n, k in zip(self.ns, self.ks):
            self.grad_sizes += [n * k, n, n]
        expect = [cluster_cols] + self.ns[:-1]
        self.ok = (len(blocks) <= 3 and expect == self.ks and all((n * k) % 2 == 0 for n, k in zip(self.ns, self.ks))
                   and all(p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda for p in self.params)
                   and int(L.lib.ococc_sir_rel_chain_fwd_floats(ctypes.byref(d), 64)) >= 0)
        self._layouts = {}

    def valid_for(self, params, rel_cs):
        return rel_cs is self.rel_cs and len(params) == len(self.params) and all(a is b for a, b in zip(params, self.params)) \
            and tuple(p.data_ptr() for p in params) == self.ptrs

    def bwd_layout(self, rows):
        hit = self._layouts.get(rows)
        if hit is None:
            ln_off, w_off = (ctypes.c_int64 * 4)(), (ctypes.c_int64 * 4)()
            tiles, slices, total = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int64()
            L.check(L.lib.ococc_sir_rel_chain_bwd_layout(ctypes.byref(self.desc), rows, ln_off, w_off, ctypes.byref(tiles),
                                                         ctypes.byref(slices), ctypes.byref(total)), 'sir_rel_chain_bwd_layout')
            if len(self._layouts) > 64:
                self._layouts.clear()
            hit = self._layouts[rows] = (list(ln_off)[:self.nl], list(w_off)[:self.nl], tiles.value, slices.value, total.value)
        return hit


def _chain_array(plans):
    arr = (_RelChainDesc * len(plans))()
    for i, p in enumerate(plans):
        arr[i] = p.desc
    return arr


def _vp(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


class _RelChains(torch.autograd.Function):
    """gates of several SIRLayers = their rel_mlps on the cluster offsets the layers share: one launch forward, one launch
    (+ the weight-gradient launch) backward, for all of them."""

    @staticmethod
    def forward(ctx, plans, f_cluster, *params):
        fc = _f32c(f_cluster)
        rows, dev = fc.shape[0], fc.device
        slabs = [torch.empty((int(L.lib.ococc_sir_rel_chain_fwd_floats(ctypes.byref(p.desc), rows)),), dtype=torch.float32,
                             device=dev) for p in plans]
        gates = [torch.empty((rows, p.ns[-1]), dtype=torch.float32, device=dev) for p in plans]
        L.check(L.lib.ococc_sir_rel_chains_fwd_f32(len(plans), _chain_array(plans), fc.data_ptr(), rows,
                                                   _vp([t.data_ptr() for t in slabs]), _vp([t.data_ptr() for t in gates]),
                                                   L.stream()), 'sir_rel_chains_fwd')
        ctx.plans = plans
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(fc, *slabs, *gates, *params)
        return tuple(gates)

    @staticmethod
    def backward(ctx, *dgates):
        plans = ctx.plans
        n = len(plans)
        t = ctx.saved_tensors
        fc, slabs, gates = t[0], t[1:1 + n], t[1 + n:1 + 2 * n]
        rows, dev = fc.shape[0], fc.device
        need = ctx.needs_input_grad
        dg = [torch.zeros_like(g) if d is None else _f32c(d) for d, g in zip(dgates, gates)]
        layouts = [p.bwd_layout(rows) for p in plans]
        bslabs = [torch.empty((lay[4],), dtype=torch.float32, device=dev) for lay in layouts]
        L.check(L.lib.ococc_sir_rel_chains_bwd_f32(n, _chain_array(plans), fc.data_ptr(), rows, _vp([s.data_ptr() for s in slabs]),
                                                   _vp([g.data_ptr() for g in gates]), _vp([d.data_ptr() for d in dg]),
                                                   _vp([s.data_ptr() for s in bslabs]), L.stream()), 'sir_rel_chains_bwd')
        grads, at = [], 2
        for p, lay, bs in zip(plans, layouts, bslabs):
            if rows == 0:
                grads += [torch.zeros_like(q) for q in p.params]
            else:
                ln_off, w_off, tiles, slices, _ = lay
                grads += _finish_param_grads(p, bs, ln_off, w_off, tiles, slices, all(need[at:at + len(p.params)]))
            at += len(p.params)
        return (None, None, *grads)


def rel_gates(layers, f_cluster):
    """[gate of every layer] for SIRLayers that share ``f_cluster`` (the layers of one SIR stack), computed by ONE launch, or
    None when that form does not apply (then every layer runs its own rel_mlp, as before)."""
    layers = list(layers)
    if not (BATCH_REL_CHAINS and NATIVE_LAYER and WHOLE_LAYER_NODE and POINT_LAYER_KERNEL and 2 <= len(layers) <= 8
            and f_cluster is not None and f_cluster.is_cuda and f_cluster.dtype == torch.float32 and not f_cluster.requires_grad
            and 0 < f_cluster.shape[0] <= REL_CHAINS_MAX_ROWS and not torch.cuda.is_current_stream_capturing()):
        return None
    plans, params = [], []
    scale = None
    for layer in layers:
        ok, rel, vfe = layer._blocks()
        if not (ok and rel and isinstance(layer, SIRLayer)):
            return None
        s = 1.0 / float(layer.rel_dist_scaler)
        rel_cs = const_tensor([s] * rel[0][0].in_features, f_cluster.device)
        if rel[0][0].in_features != f_cluster.shape[1]:
            return None
        ps = []
        for lin, norm, _ in rel:
            ps += [lin.weight, norm.weight, norm.bias]
        plan = _REL_PLANS.get(layer)
        if plan is None or not plan.valid_for(ps, rel_cs):
            plan = _REL_PLANS[layer] = _RelPlan(rel, f_cluster.shape[1], rel_cs)
        if not plan.ok or (plans and plan.nl != plans[0].nl):
            return None
        plans.append(plan)
        params += ps
    # the weight fragments of the whole stack (the rel plans, and the layers' vfe plans once they exist) in shared launches
    vfe_plans = [_VFE_PLANS[layer] for layer in layers if layer in _VFE_PLANS]
    refresh_plans([p.pack for p in plans] + [p.pack for p in vfe_plans], backward=torch.is_grad_enabled())
    return list(_RelChains.apply(plans, f_cluster, *params))


class _SirLayerFn(torch.autograd.Function):
    """SIRLayer._forward_fused as ONE node: forward = its point_mlp launches in order, backward = their backward
    launches in reverse (the gradient of a layer's input is the next call's dy; the gradient of the maxima a layer
    gathered joins the one that arrives from outside).  ``spec`` = (rel blocks, vfe blocks, acts, eps, rel scale
    constant, column scale constant, bscale, with cluster centre, shortcut); ``params`` = (weight, ln weight, ln bias)
    per block, rel_mlp first."""

    @staticmethod
    def forward(ctx, spec, features, f_cluster, inv, G, *params):
        nr, nv, acts, epss, rel_cs, col, bscale, with_cc, shortcut = spec
        feats, fc = _f32c(features), _f32c(f_cluster)
        ws = [params[3 * i] for i in range(nr + nv)]
        gs = [_f32c(params[3 * i + 1]) for i in range(nr + nv)]
        bs = [_f32c(params[3 * i + 2]) for i in range(nr + nv)]
        wfs = [pack_weight(w.detach(), w) for w in ws]
        ys, ms = [], []
        x = fc
        for j in range(nr):   # gate = rel_mlp(f_cluster / rel_dist_scaler)
            x, _ = layer_forward(x, None, None, None, wfs[j], ws[j].shape[0], gs[j], bs[j], rel_cs if j == 0 else None, None,
                                 1.0, epss[j], acts[j], False, 0)
            ys.append(x)
        gate = x if nr else None
        y = None
        for i in range(nv):
            q = nr + i
            if i == 0:
                y, m = layer_forward(feats, gate, fc if with_cc else None, None, wfs[q], ws[q].shape[0], gs[q], bs[q], col, inv,
                                     bscale, epss[q], acts[q], True, G)
            else:
                y, m = layer_forward(y, None, None, ms[-1], wfs[q], ws[q].shape[0], gs[q], bs[q], None, inv, 1.0, epss[q],
                                     acts[q], True, G)
            ys.append(y)
            ms.append(m)
        ctx.spec = spec
        ctx.G = G
        ctx.ln_params = [(params[3 * i + 1], params[3 * i + 2]) for i in range(nr + nv)]
        ctx.w_params = [params[3 * i] for i in range(nr + nv)]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(feats, fc, inv, *ws, *gs, *bs, *wfs, *ys, *ms)
        out = y + feats[:, 3:] if shortcut else y
        groups = torch.cat(ms, dim=1) if nv > 1 else ms[0]
        return out, groups

    @staticmethod
    def backward(ctx, dy, dM):
        nr, nv, acts, epss, rel_cs, col, bscale, with_cc, shortcut = ctx.spec
        G, nl = ctx.G, nr + nv
        t = ctx.saved_tensors
        feats, fc, inv = t[0], t[1], t[2]
        ws, gs, bs, wfs, ys = (t[3 + k * nl: 3 + (k + 1) * nl] for k in range(5))
        ms = t[3 + 5 * nl:]
        need = ctx.needs_input_grad
        rows = feats.shape[0]
        if dy is None:
            dy = torch.zeros((rows, ys[-1].shape[1]), dtype=torch.float32, device=feats.device)
        dy_out = dy = _f32c(dy)
        grads = [None] * (3 * nl)

        def param_grads(q, dw, lnp, tiles):
            grads[3 * q] = dw
            lw, lb = ctx.ln_params[q]
            dg, db = ln_param_grads(lw, lb, lnp, tiles, ws[q].shape[0], need[5 + 3 * q + 1] and need[5 + 3 * q + 2])
            grads[3 * q + 1], grads[3 * q + 2] = dg, db

        off = [0]
        for m in ms:
            off.append(off[-1] + m.shape[1])
        carry, dfeat, dgate = None, None, None
        for i in reversed(range(nv)):
            q = nr + i
            dm = None if dM is None else dM[:, off[i]:off[i + 1]]
            if carry is not None:
                dm = carry if dm is None else torch.add(dm, carry)
            if dm is not None:
                dm = _f32c(dm)
            if i == 0:
                gate = ys[nr - 1] if nr else None
                da, dmul, _, _, dw, lnp, tiles = layer_backward(
                    feats, gate, fc if with_cc else None, None, ws[q], wfs[q], gs[q], bs[q], col, inv, ys[q], ms[i], bscale,
                    epss[q], acts[q], G, dy, dm, need[1], True, False, False, need[5 + 3 * q], ctx.w_params[q])
                dfeat, dgate = da, dmul
            else:
                da, _, _, dv, dw, lnp, tiles = layer_backward(
                    ys[q - 1], None, None, ms[i - 1], ws[q], wfs[q], gs[q], bs[q], None, inv, ys[q], ms[i], 1.0, epss[q],
                    acts[q], G, dy, dm, True, False, False, True, need[5 + 3 * q], ctx.w_params[q])
                dy, carry = da, dv
            param_grads(q, dw, lnp, tiles)
        if shortcut and dfeat is not None:
            dfeat[:, 3:] += dy_out
        for j in reversed(range(nr)):
            x_in = ys[j - 1] if j > 0 else fc
            da, _, _, _, dw, lnp, tiles = layer_backward(
                x_in, None, None, None, ws[j], wfs[j], gs[j], bs[j], rel_cs if j == 0 else None, None, ys[j], None, 1.0,
                epss[j], acts[j], 0, dgate, None, j > 0, False, False, False, need[5 + 3 * j], ctx.w_params[j])
            dgate = da
            param_grads(j, dw, lnp, tiles)
        return (None, dfeat, None, None, None, *grads)


class DynamicVFELayerV2(nn.Module):
    """Linear(no bias) -> norm -> act (utils.py:147-189)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 act='relu', dropout=0.0):
        super().__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = Linear(in_channels, out_channels, bias=False)
        self.act = get_activation_layer(act, out_channels)
        self.norm, self.act = fuse_norm_act(self.norm, self.act)
        self.dropout = nn.Dropout(p=dropout) if dropout > 0 else None

    def forward(self, inputs):
        if self.dropout is not None:
            inputs = self.dropout(inputs)
        return self.act(self.norm(self.linear(inputs)))


@VOXEL_ENCODERS.register_module()
class SIRLayer(nn.Module):
    """voxel_encoder.py:686-832.  The reference derives from DynamicVFE only to inherit flags;
    the members it actually uses are kept, with the same names."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_rel_mlp=True, rel_mlp_hidden_dims=[16, ], rel_mlp_in_channel=3,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4),
                 point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=1.0, with_shortcut=True,
                 xyz_normalizer=[1.0, 1.0, 1.0], act='relu', dropout=0.0):
        super().__init__()
        assert len(feat_channels) > 0
        raw_in_channels = in_channels
        # DynamicVFE.__init__ (voxel_encoder.py:136-143) widens in_channels for the decorations
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 3
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.rel_dist_scaler = rel_dist_scaler
        self.mode = mode
        self.with_shortcut = with_shortcut
        self._with_rel_mlp = with_rel_mlp
        self.xyz_normalizer = xyz_normalizer
        if with_rel_mlp:
            # the reference appends to the caller's list (voxel_encoder.py:733); the config
            # loader hands every block its own copy, and so do we
            rel_mlp_hidden_dims = list(rel_mlp_hidden_dims) + [raw_in_channels]  # 'not self.in_channels'
            self.rel_mlp = build_mlp(rel_mlp_in_channel, rel_mlp_hidden_dims, norm_cfg, act=act)
        feat_channels = [self.in_channels] + list(feat_channels)
        vfe_layers = []
        for i in range(len(feat_channels) - 1):
            in_filters = feat_channels[i]
            if i > 0:
                in_filters *= 2
            vfe_layers.append(DynamicVFELayerV2(in_filters, feat_channels[i + 1], norm_cfg, act=act,
                                                dropout=dropout))
        self.vfe_layers = nn.ModuleList(vfe_layers)
        self.num_vfe = len(vfe_layers)

    # ---- forward ------------------------------------------------------------------------------------------------
    # What the layer computes (voxel_encoder.py:764-832), per point p of group g = inv[p]:
    #   c_p   = f_cluster_p / rel_dist_scaler          (given, or xyz_p - mean of the group's xyz)
    #   x_p   = [ (xyz_p / xyz_normalizer, rest_p) * rel_mlp(c_p) | c_p / 10 if with_cluster_center | |xyz_p| if with_distance ]
    #   y0_p  = vfe_0(x_p),  m0_g = max_{p in g} y0_p,   y1_p = vfe_1([y0_p | m0_g]),  m1_g = max y1_p, ...
    # returned: the last y (+ the raw non-xyz input columns when the widths agree) and [m0 | m1 | ...].
    # Two realisations: `_forward_fused` -- every Linear -> LN -> act (+ the max) of the chain is ONE launch of
    # csrc/point_mlp.hip, the concatenations / products / gather-backs happen while the kernel assembles its input
    # rows -- and `_forward_ops`, the same chain from separate operators for the options the kernel does not cover.
    def _groups(self, coors, inv, group_coors):
        if inv is None:
            group_coors, inv = unique_with_inverse(coors)
        return inv, group_coors

    def _cluster_offsets(self, xyz, f_cluster, inv, num_groups):
        if f_cluster is not None:
            return f_cluster
        centre = segment_reduce(xyz.float(), inv, num_groups, 'mean')
        return xyz - gather_rows(centre, inv)

    def _fusable(self):
        from .norm import LayerNorm
        if not POINT_LAYER_KERNEL or self.mode != 'max' or self._with_distance:
            return False
        blocks = [(v.linear, v.norm, v.act, v.dropout) for v in self.vfe_layers]
        if self._with_rel_mlp:
            blocks += [(b[0], b[1], b[2], b[3] if len(b) > 3 else None) for b in self.rel_mlp]
        for lin, norm, act, drop in blocks:
            if not (isinstance(norm, LayerNorm) and norm.fused_act in ('gelu', 'none') and isinstance(act, (nn.Identity, nn.ReLU))
                    and lin.bias is None and lin.in_features <= 256 and lin.out_features <= 144):
                return False
            if drop is not None and self.training and getattr(drop, 'p', 0) > 0:
                return False
        return True

    @staticmethod
    def _act_of(norm, act):
        return 'gelu' if norm.fused_act == 'gelu' else ('relu' if isinstance(act, nn.ReLU) else 'none')

    def _blocks(self):
        """(Linear, norm, act name) of every block, rel_mlp first -- and whether the point_mlp kernels cover them; per
        training mode (dropout only counts while training), kept until the kernel switch is flipped."""
        key = (self.training, POINT_LAYER_KERNEL)
        cache = self.__dict__.setdefault('_block_cache', {})
        if key not in cache:
            ok = self._fusable()
            rel = [(b[0], b[1], self._act_of(b[1], b[2])) for b in self.rel_mlp] if (ok and self._with_rel_mlp) else []
            vfe = [(v.linear, v.norm, self._act_of(v.norm, v.act)) for v in self.vfe_layers] if ok else []
            cache[key] = (ok, rel, vfe)
        return cache[key]

    def _forward_fused(self, features, f_cluster, inv, num_groups, shortcut=False, gate=None):
        dev = features.device
        _, rel, vfe = self._blocks()
        raw = self.in_channels - 3 * (self._with_cluster_center + self._with_voxel_center)   # columns of `features`
        scale = 1.0 / float(self.rel_dist_scaler)
        col = const_tensor([1.0 / v for v in self.xyz_normalizer] + [1.0] * (raw - 3), dev)
        rel_cs = const_tensor([scale] * rel[0][0].in_features, dev) if rel else None
        inv_in = inv
        if inv.dtype != torch.int32:   # (the blocks of a SIR stack share one inverse: converted once, kept on the tensor)
            i32 = getattr(inv, '_ococc_i32', None)
            if i32 is None:
                i32 = inv.to(torch.int32)
                inv._ococc_i32 = i32
            inv = i32
        blocks = rel + vfe
        whole = (WHOLE_LAYER_NODE and features.dtype == torch.float32 and f_cluster.dtype == torch.float32
                 and not f_cluster.requires_grad and all(lin.weight.dtype == torch.float32 for lin, _, _ in blocks))
        if gate is not None:   # the rel_mlp ran elsewhere (rel_gates: all layers of the stack in one launch): vfe blocks only
            plan = _VFE_PLANS.get(self)
            params = []
            for lin, norm, _ in vfe:
                params += [lin.weight, norm.weight, norm.bias]
            if plan is None or not plan.valid_for(params) or plan.consts[1] is not col:
                plan = _VFE_PLANS[self] = _NativePlan(vfe, 0, features.shape[1], f_cluster.shape[1], self._with_cluster_center,
                                                      scale / 10.0, None, col)
            if not (whole and plan.ok and gate.shape == features.shape and gate.dtype == torch.float32):
                # (rel_gates() hands out gates before the layers see their features: half-precision features under autocast,
                # an unusual vfe shape -- the same arithmetic operator by operator, with the gate as an input)
                y, groups = self._forward_ops(features, f_cluster, inv_in, num_groups, gate)
                return y, groups, False
            plan.pack.refresh(backward=torch.is_grad_enabled())
            y, groups = _SirLayerNative.apply(plan, bool(shortcut), features, f_cluster, inv, int(num_groups), gate, *params)
            return y, groups, shortcut
        if whole and NATIVE_LAYER and len(blocks) <= 8 and not torch.cuda.is_current_stream_capturing():
            plan = _NATIVE_PLANS.get(self)
            params = []
            for lin, norm, _ in blocks:
                params += [lin.weight, norm.weight, norm.bias]
            if plan is None or not plan.valid_for(params) or plan.consts[0] is not rel_cs or plan.consts[1] is not col:
                plan = _NATIVE_PLANS[self] = _NativePlan(blocks, len(rel), features.shape[1], f_cluster.shape[1],
                                                         self._with_cluster_center, scale / 10.0, rel_cs, col)
            if plan.ok and all(p.is_cuda for p in params):
                plan.pack.refresh(backward=torch.is_grad_enabled())
                y, groups = _SirLayerNative.apply(plan, bool(shortcut), features, f_cluster, inv, int(num_groups), None, *params)
                return y, groups, shortcut
        # all Linears of this layer (and their transposes, when a backward pass will follow) packed in one launch
        lins = [lin.weight for lin, _, _ in blocks]
        if all(w.dtype == torch.float32 and w.is_cuda for w in lins):
            plan = _PACK_PLANS.get(self)
            if plan is None or not plan.valid_for(lins):
                plan = _PACK_PLANS[self] = PackPlan(lins)
            plan.refresh(backward=torch.is_grad_enabled())
        else:
            prepack(lins, backward=torch.is_grad_enabled())
        if whole:
            spec = (len(rel), len(vfe), tuple(a for _, _, a in blocks), tuple(float(n.eps) for _, n, _ in blocks), rel_cs, col,
                    scale / 10.0, bool(self._with_cluster_center), bool(shortcut))
            params = []
            for lin, norm, _ in blocks:
                params += [lin.weight, norm.weight, norm.bias]
            y, groups = _SirLayerFn.apply(spec, features, f_cluster, inv, int(num_groups), *params)
            return y, groups, shortcut
        gate = None
        if rel:   # gate = rel_mlp(f_cluster / rel_dist_scaler), one launch per Linear -> LN -> act
            gate = f_cluster
            for i, (lin, norm, act) in enumerate(rel):
                gate = point_layer(gate, lin.weight, norm.weight, norm.bias, norm.eps, act, colscale=rel_cs if i == 0 else None)
        extra = f_cluster if self._with_cluster_center else None
        maxima = []
        y = None
        for i, (lin, norm, act) in enumerate(vfe):
            if i == 0:
                y, m = point_layer(features, lin.weight, norm.weight, norm.bias, norm.eps, act, mul=gate,
                                   colscale=col, b=extra, bscale=scale / 10.0, inv=inv, num_segments=num_groups, seg_max=True)
            else:
                y, m = point_layer(y, lin.weight, norm.weight, norm.bias, norm.eps, act, v=maxima[-1],
                                   inv=inv, num_segments=num_groups, seg_max=True)
            maxima.append(m)
        return y, torch.cat(maxima, dim=1), False

    def _forward_ops(self, features, f_cluster, inv, num_groups, gate=None):
        xyz = features[:, :3]
        scaled = f_cluster / self.rel_dist_scaler
        head = torch.cat([xyz / const_tensor(self.xyz_normalizer, features.device, features.dtype)[None, :],
                          features[:, 3:]], dim=1)
        if self._with_rel_mlp:
            head = head * (self.rel_mlp(scaled) if gate is None else gate)
        parts = [head]
        if self._with_cluster_center:
            parts.append(scaled / 10.0)
        if self._with_distance:
            parts.append(torch.norm(xyz, 2, 1, keepdim=True))
        x = torch.cat(parts, dim=-1)
        maxima = []
        for i, vfe in enumerate(self.vfe_layers):
            y = vfe(x)
            maxima.append(segment_reduce(y.float(), inv, num_groups, 'mean' if self.mode == 'avg' else self.mode))
            if i + 1 < len(self.vfe_layers):
                x = torch.cat([y, gather_rows(maxima[-1], inv)], dim=1)
        return y, torch.cat(maxima, dim=1)

    def forward(self, features, coors, f_cluster=None, points=None, img_feats=None, img_metas=None,
                return_inv=False, return_both=False, unq_inv_once=None, new_coors_once=None, gate=None):
        # ``gate`` (not in the reference's signature): this layer's rel_mlp output when the caller computed the gates of a
        # whole stack at once (rel_gates); None = the layer runs its own rel_mlp
        inv, group_coors = self._groups(coors, unq_inv_once, new_coors_once)
        num_groups = group_coors.size(0)
        f_cluster = self._cluster_offsets(features[:, :3], f_cluster, inv, num_groups)
        want_points = return_both or self.return_point_feats
        assert gate is None or self._blocks()[0]
        if self._blocks()[0] and features.shape[0] <= POINT_LAYER_MAX_ROWS:
            n_out = self.vfe_layers[-1].linear.out_features
            shortcut = bool(want_points and self.with_shortcut and n_out == features.shape[1] - 3)
            point_feats, group_feats, shortcut_done = self._forward_fused(features, f_cluster, inv, num_groups, shortcut, gate)
        else:
            point_feats, group_feats = self._forward_ops(features, f_cluster, inv, num_groups, gate)
            shortcut_done = False
        if want_points:
            if not shortcut_done and self.with_shortcut and point_feats.shape[1] == features.shape[1] - 3:
                point_feats = point_feats + features[:, 3:]
            return (point_feats, group_feats, group_coors) if return_both else (point_feats, group_feats)
        return (group_feats, group_coors, inv) if return_inv else (group_feats, group_coors)


@BACKBONES.register_module()
class SIR(nn.Module):
    """Stack of SIRLayers (backbones/sir.py:16-88)."""

    def __init__(self, num_blocks=5, in_channels=[], feat_channels=[], rel_mlp_hidden_dims=[],
                 with_rel_mlp=True, with_distance=False, with_cluster_center=False,
                 norm_cfg=dict(type='LN', eps=1e-3), mode='max', xyz_normalizer=[1.0, 1.0, 1.0],
                 act='relu', dropout=0, unique_once=False):
        super().__init__()
        self.num_blocks = num_blocks
        self.unique_once = unique_once
        block_list = []
        for i in range(num_blocks):
            block_list.append(SIRLayer(
                in_channels=in_channels[i], feat_channels=feat_channels[i], with_distance=with_distance,
                with_cluster_center=with_cluster_center, with_rel_mlp=with_rel_mlp,
                rel_mlp_hidden_dims=rel_mlp_hidden_dims[i], with_voxel_center=False,
                voxel_size=[0.1, 0.1, 0.1], point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4],
                norm_cfg=norm_cfg, mode=mode, fusion_layer=None,
                return_point_feats=i != num_blocks - 1, return_inv=False, rel_dist_scaler=10.0,
                xyz_normalizer=xyz_normalizer, act=act, dropout=dropout))
        self.block_list = nn.ModuleList(block_list)

    def forward(self, points, features, coors, f_cluster=None, dims=None):
        """points [M, 3], features [M, C] -> (point features of the last block, [group maxima of all blocks], group coors);
        the groups are found once when ``unique_once`` (backbones/sir.py:67-88)."""
        group_coors, inv = unique_with_inverse(coors, dims) if self.unique_once else (None, None)
        if f_cluster is None and inv is not None and not points.requires_grad:
            # every block would derive the same offsets from the same xyz columns and the same groups (voxel_encoder.py:
            # 777-781) -- and, taken from the concatenated input, they would formally require a gradient nobody uses
            centre = segment_reduce(points.float(), inv, group_coors.size(0), 'mean')
            f_cluster = points - gather_rows(centre, inv)
        feats, per_block = features, []
        last = len(self.block_list) - 1
        # the gates of all blocks from the offsets they share, in one launch (None: every block runs its own rel_mlp)
        gates = rel_gates(self.block_list, f_cluster) if (f_cluster is not None and inv is not None) else None
        for i, block in enumerate(self.block_list):
            out = block(torch.cat([points, feats], 1), coors, f_cluster, return_both=(i == last), unq_inv_once=inv,
                        new_coors_once=group_coors, **({} if gates is None else {'gate': gates[i]}))
            feats = out[0]
            per_block.append(out[1])
        return feats, torch.cat(per_block, dim=1), out[2]
