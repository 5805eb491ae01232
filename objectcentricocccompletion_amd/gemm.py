"""Matrix products of the f32 parts of the model (temporal transformer, the heads' MLPs, the SIR layers' Linears above the
fused per-point kernel's row limit) with bf16 OPERANDS on the matrix cores and f32 accumulation -- opt-in.

The reference computes these in f32 (nn.Linear / nn.MultiheadAttention, mmdet3d/models/occ/layers.py:35-87,
ococc_bbox_head.py:849-908).  On MI355X an f32 GEMM runs on the vector ALUs' rate (157 TFLOP/s peak); at 64 tracklets per
GPU these products are 17 of the 56 ms step (profiles/r05_ococcnet_b64_kernel_stats.csv: the `Cijk_..._S_B_` kernels).
With GEMM_DTYPE = torch.bfloat16 both operands are rounded to bf16 (what north_star allows for features: 1e-3), products
are exact, sums f32; outputs, biases, LayerNorm, softmax and the residual stream stay f32.  Gradients the same way
(dX = dY W, dW = dY^T X with bf16 operands, f32 sums).  Default: off (f32 as the reference; the goldens of the imported
reference are held at 1e-4 in that mode); OCOCC_GEMM_DTYPE=bf16 or gemm.GEMM_DTYPE = torch.bfloat16 switches it on.
"""
import os

import torch
import torch.nn.functional as F

GEMM_DTYPE = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16}.get(os.environ.get('OCOCC_GEMM_DTYPE', '').lower())
# EMULATE: f32 GEMMs on operands ROUNDED to bf16, forward and backward -- the same numbers as the bf16 products up to the order of the f32 sums
# (tests: the mixed path computes what it says; what bf16 operands cost against the f32 reference is then one subtraction)
EMULATE = False
_TALL = 16384   # rows from which a weight gradient is contracted in slices (fused_mlp.wgrad_rows_bf16)


def _r(t, dt):
    """operand of a product: rounded to ``dt``; EMULATE: rounded, then back in f32 (the product then runs as an f32 GEMM)"""
    t = t.to(dt)
    return t.float() if EMULATE else t


def _mm(a, b):
    return torch.mm(a, b) if a.dtype == torch.float32 else torch.mm(a, b, out_dtype=torch.float32)


def _bmm(a, b):
    return torch.bmm(a, b) if a.dtype == torch.float32 else torch.bmm(a, b, out_dtype=torch.float32)


class _MixedLinear(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, w, b):
        dt = GEMM_DTYPE
        x2 = x.reshape(-1, x.shape[-1])
        x16, w16 = _r(x2, dt), _r(w, dt)
        y = _mm(x16, w16.t())
        if b is not None:
            y = y + b.float()
        ctx.save_for_backward(x16, w16)
        ctx.meta = (x.shape, x.dtype, w.dtype, None if b is None else b.dtype)
        return y.view(*x.shape[:-1], w.shape[0]).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x16, w16 = ctx.saved_tensors
        shape, xdt, wdt, bdt = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1])
        dy16 = _r(dy2, GEMM_DTYPE)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _mm(dy16, w16).view(shape).to(xdt)
        if ctx.needs_input_grad[1]:
            if dy16.shape[0] >= _TALL and dy16.dtype != torch.float32:
                from .occ.fused_mlp import wgrad_rows_bf16
                dw = wgrad_rows_bf16(dy16, x16).to(wdt)
            else:
                dw = _mm(dy16.t(), x16).to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            db = dy2.sum(0).to(bdt)
        return dx, dw, db


class _MixedBmm(torch.autograd.Function):
    """a [B, M, K] @ b [B, K, N] with bf16 operands"""

    @staticmethod
    def forward(ctx, a, b):
        a16, b16 = _r(a, GEMM_DTYPE), _r(b, GEMM_DTYPE)
        ctx.save_for_backward(a16, b16)
        ctx.dts = (a.dtype, b.dtype)
        return _bmm(a16, b16).to(a.dtype)

    @staticmethod
    def backward(ctx, dy):
        a16, b16 = ctx.saved_tensors
        dy16 = _r(dy, GEMM_DTYPE)
        da = _bmm(dy16, b16.transpose(1, 2)).to(ctx.dts[0]) if ctx.needs_input_grad[0] else None
        db = _bmm(a16.transpose(1, 2), dy16).to(ctx.dts[1]) if ctx.needs_input_grad[1] else None
        return da, db


def _mixed(x):
    return GEMM_DTYPE is not None and x.is_cuda and x.dtype == torch.float32


# ---------------------------------------------------------------------------------------------------------------------
# f32 products as ONE bf16 GEMM over three-way split operands (round 6; csrc/split3.hip has the derivation and the measured
# accuracy: 4.5e-6 of f64 where the f32 GEMM has 7e-7 -- the goldens of the imported reference hold at their f32 tolerances).
# The default for the f32 Linears from SPLIT3_MIN_ROWS rows on, where the f32 library GEMM is bound by the f32 matrix
# instructions (tools/probe/gemm_x3_probe.py, MI355X: 3072 x 1536 at 2 048 rows 150 -> 53 us, at 512 rows 48 -> 26 us; at 128
# rows both are one trip through the weights).  A product becomes four launches (two splits, the GEMM, the bias) where the
# library needs one, ~9 us each: whole configs[2] step at 64 tracklets (2 048 RoIs) 55.3 -> 53.9 ms, at 16 tracklets (512
# RoIs) 20.0 -> 21.4 ms -- hence 1 024 rows.  y = x w^T, dx = dy w, dw = dy^T x each one GEMM; the weight's two operands
# ([N, 3K] for y, [3N, K] for dx) are made by one pass per optimizer step and kept per parameter.
SPLIT3 = os.environ.get('OCOCC_GEMM_SPLIT3', '1') == '1'
SPLIT3_MIN_ROWS = int(os.environ.get('OCOCC_GEMM_SPLIT3_MIN_ROWS', '1024'))
SPLIT3_MIN_WORK = 1 << 29          # rows x N x K below which the product is a launch's worth either way
_HHL, _HLH = 0, 1                   # (hi, hi, lo) / (hi, lo, hi): one operand of a product takes the one, the other the other
# parameter (the BASE tensor of a row slice) -> {(data_ptr, N, K, row stride): [version, cat [N, 3K], stack [3N, K]]}, keyed by the
# tensor OBJECT and weakly: an address is reused by the caching allocator, a dead parameter's operands must not be
from torch.utils.weak import WeakIdKeyDictionary   # noqa: E402
_w_operands = WeakIdKeyDictionary()


def split3(t, cat=None, stack=None):
    """The three-part bf16 operands of the f32 matrix ``t`` [R, C] (ococc_split3_bf16): ``cat`` / ``stack`` name the pattern
    (_HHL / _HLH) of the [R, 3C] / [3R, C] form wanted, None = not wanted.  One pass over ``t``."""
    from . import _lib as L
    assert t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1
    R, C = t.shape
    oc = torch.empty((R, 3 * C), dtype=torch.bfloat16, device=t.device) if cat is not None else None
    os_ = torch.empty((3 * R, C), dtype=torch.bfloat16, device=t.device) if stack is not None else None
    L.check(L.lib.ococc_split3_bf16(t.data_ptr(), R, C, t.stride(0), L.ptr(oc), cat or 0, L.ptr(os_), stack or 0, L.stream()),
            'split3')
    return oc, os_


def _weight_operands(w, want):
    """The operand of the weight ``w`` [N, K] (a parameter or a row slice of one: in_proj_weight[:2E]) in pattern _HLH:
    ``want`` = 'cat' ([N, 3K], for y = x w^T) or 'stack' ([3N, K], for dx = dy w).  Eagerly both forms are made by one pass
    per VALUE of the parameter (its version counter moves with every optimizer step, also our AdamW's: optim.py) and kept.
    While a HIP graph is being captured (heads.graphed_call: the temporal transformer and the head's tail replay as graph
    pairs) nothing is looked up: the split is recorded into the graph, into buffers of the graph's own pool -- a replay
    then splits the weights of ITS step, and the buffers live as long as the graph."""
    if torch.cuda.is_current_stream_capturing():
        cat, stack = split3(w.detach(), cat=_HLH if want == 'cat' else None, stack=_HLH if want == 'stack' else None)
        return cat if want == 'cat' else stack
    base = w._base if w._base is not None else w
    table = _w_operands.get(base)
    if table is None:
        table = _w_operands[base] = {}
    key = (w.data_ptr(), w.shape[0], w.shape[1], w.stride(0))
    hit = table.get(key)
    if hit is None or hit[0] != w._version or hit[1].device != w.device:
        if len(table) > 16:
            table.clear()
        cat, stack = split3(w.detach(), cat=_HLH, stack=_HLH)
        hit = table[key] = [w._version, cat, stack]
    return hit[1] if want == 'cat' else hit[2]


class _Split3Linear(torch.autograd.Function):
    """y = x w^T (+ b) for f32 x [M, K], w [N, K] through three-way split bf16 operands; f32 in, f32 out."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need_dw = ctx.needs_input_grad[1]
        xc, xs = split3(x2, cat=_HHL, stack=_HLH if need_dw else None)
        wc = _weight_operands(w, 'cat')
        y = torch.mm(xc, wc.t(), out_dtype=torch.float32)
        if b is not None:
            y = y + b
        ctx.save_for_backward(w, *(() if xs is None else (xs,)))
        ctx.meta = (x.shape, b is not None)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        w = ctx.saved_tensors[0]
        xs = ctx.saved_tensors[1] if len(ctx.saved_tensors) > 1 else None
        shape, has_b = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous() or dy2.dtype != torch.float32:
            dy2 = dy2.float().contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dyc, dys = split3(dy2, cat=_HHL if need_dx else None, stack=_HHL if need_dw else None)
        dx = dw = db = None
        if need_dx:
            ws = _weight_operands(w, 'stack')
            dx = torch.mm(dyc, ws, out_dtype=torch.float32).view(shape)
        if need_dw:
            dw = torch.mm(dys.t(), xs, out_dtype=torch.float32)
        if has_b and ctx.needs_input_grad[2]:
            db = dy2.sum(0)
        return dx, dw, db


def _split3_ok(x, w, b):
    if not (SPLIT3 and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.dim() == 2
            and (b is None or b.dtype == torch.float32)):
        return False
    K = x.shape[-1]
    M = x.numel() // max(K, 1)
    N = w.shape[0]
    return (M >= SPLIT3_MIN_ROWS and K % 8 == 0 and N % 8 == 0 and w.stride(1) == 1 and w.stride(0) % 4 == 0
            and M * N * K >= SPLIT3_MIN_WORK and w.data_ptr() % 16 == 0)


def linear(x, w, b=None):
    """F.linear(x, w, b).  f32 on the device: from SPLIT3_MIN_ROWS rows on one bf16 GEMM over three-way split operands
    (f32-level accuracy, see above); with GEMM_DTYPE set: plain bf16 operands, f32 accumulation (see the module docstring)."""
    if _mixed(x) and w.dtype == torch.float32:
        return _MixedLinear.apply(x, w, b)
    if _split3_ok(x, w, b):
        return _Split3Linear.apply(x, w, b)
    return F.linear(x, w, b)


def bmm(a, b):
    if _mixed(a) and b.dtype == torch.float32:
        return _MixedBmm.apply(a, b)
    return torch.bmm(a, b)
