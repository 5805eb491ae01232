"""nn.Linear for the per-point layers of the RoI encoder (same parameters, same state-dict keys).

The SIR layers and their MLPs (voxel_encoder.py:686-832, sst_ops.py:333-360) apply Linear(16..144 -> 3..144) to every
point of the batch: 1.3e5 rows at 64 tracklets.  Their weight gradient dW = dY^T X is then a GEMM with a tiny output and
a contraction 1e5 long; the library runs it as one 32x32 macro-tile per output block (measured on MI355X: 340-380 us
per layer, 47 such GEMMs = 19 ms of a 157 ms step).  Here the rows are cut into slices of 4096, the slices contracted
as ONE batched GEMM and the partial products summed (fixed order): 35-65 us (tools/probe/tall_wgrad.py).  Forward and
input gradient are the ordinary GEMMs."""
import torch
import torch.nn.functional as F
from torch import nn

import os
TALL_ROWS = int(os.environ.get('OCOCC_TALL_ROWS', 16384))   # from this many rows on (and <= 256 features) the sliced weight gradient is used
_SLICE = 4096


def sliced_wgrad(gy, x, rows=None):
    """dY^T X -> [out, in], contraction over the rows in slices of ``rows`` (+ a remainder).  Default slice height: 4096
    rows, less for inputs of a few 1e4 rows so that the batched GEMM still has ~64 slices to spread over the chip (the
    library runs each slice's [out, in] product as a handful of 32 x 32 macro tiles: 8 slices of 33 k rows took 49 us)."""
    n, cout = gy.shape
    cin = x.shape[1]
    if rows is None:
        rows = min(_SLICE, max(256, (n // 64) // 256 * 256))
    s = n // rows
    out = (gy[:s * rows].view(s, rows, cout).transpose(1, 2) @ x[:s * rows].view(s, rows, cin)).sum(0)
    if s * rows < n:
        out = out + gy[s * rows:].t() @ x[s * rows:]
    return out


class _TallLinear(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = sliced_wgrad(gy, x) if ctx.needs_input_grad[1] else None
        gb = gy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


class _TallAddmm(torch.autograd.Function):
    """base + x @ w^T for 1e5..1e6 rows and a small weight (the per-query half of the occupancy decoder's first layer:
    60 positional-encoding channels -> 512): the weight gradient is the row-sliced batched GEMM above (the library
    ran it as ONE 32x32 macro-tile over a contraction of 1 M: 2.3 ms)."""

    @staticmethod
    def forward(ctx, base, x, w):
        ctx.save_for_backward(x, w)
        return torch.addmm(base, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ w if ctx.needs_input_grad[1] else None
        gw = sliced_wgrad(gy, x).to(w.dtype) if ctx.needs_input_grad[2] else None
        return (gy if ctx.needs_input_grad[0] else None), gx, gw


def tall_addmm(base, x, w):
    """torch.addmm(base, x, w.t()) with the sliced weight gradient when there are many rows."""
    if x.dim() == 2 and x.size(0) >= TALL_ROWS and x.dtype == w.dtype == base.dtype and torch.is_grad_enabled():
        return _TallAddmm.apply(base, x.contiguous(), w)
    return torch.addmm(base, x, w.t())


class Linear(nn.Linear):

    def forward(self, x):
        from . import gemm
        if gemm.GEMM_DTYPE is not None and x.is_cuda and x.dtype == torch.float32 and min(self.in_features, self.out_features) >= 16:
            return gemm.linear(x, self.weight, self.bias)   # bf16 operands on the matrix cores, f32 sums (opt-in: gemm.py)
        if (x.dim() == 2 and x.size(0) >= TALL_ROWS and max(self.in_features, self.out_features) <= 256
                and x.dtype == torch.float32 and x.is_contiguous() and torch.is_grad_enabled()):
            return _TallLinear.apply(x, self.weight, self.bias)
        if min(self.in_features, self.out_features) >= 256 and gemm._split3_ok(x, self.weight, self.bias):
            # the RoI-level MLPs (ococc_bbox_head.py:116-193: 3072 -> 2048 -> 2048 -> 1536, ...) from a few hundred RoIs on:
            # one bf16 GEMM over three-way split operands, f32-level accuracy (gemm.py)
            return gemm.linear(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
