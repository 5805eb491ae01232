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


def linear(x, w, b=None):
    """F.linear(x, w, b); with GEMM_DTYPE set: bf16 operands, f32 accumulation (see the module docstring)"""
    if _mixed(x) and w.dtype == torch.float32:
        return _MixedLinear.apply(x, w, b)
    return F.linear(x, w, b)


def bmm(a, b):
    if _mixed(a) and b.dtype == torch.float32:
        return _MixedBmm.apply(a, b)
    return torch.bmm(a, b)
