"""LayerNorm (+ fused GELU) on feature rows, HIP-backed.

Covers the norm/activation layers the reference builds with
build_norm_layer(dict(type='LN', eps=1e-3), C) + nn.GELU() inside
make_sparse_convmodule (mmdet3d/ops/sparse_block.py:216-289) and build_mlp
(mmdet3d/ops/sst/sst_ops.py:333-360).  Parameter names (weight, bias) are those of
nn.LayerNorm so reference checkpoints load.  Kernels: ococc_layernorm_act_fwd/_bwd.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib as L
from .registry import NORM_LAYERS


class _LayerNormAct(Function):

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act):
        L.require_device(x, weight, bias)
        shape = x.shape
        c = shape[-1]
        x2 = x.reshape(-1, c).contiguous()
        n = x2.size(0)
        dt = L.dtype_code(x2.dtype)
        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        y = torch.empty_like(x2)
        stats = torch.empty((n, 2), dtype=torch.float32, device=x2.device)
        L.check(L.lib.ococc_layernorm_act_fwd(L.ptr(x2), n, c, L.ptr(w32), L.ptr(b32), float(eps),
                                              int(act), L.ptr(y), L.ptr(stats), dt, L.stream()),
                'layernorm_act_fwd')
        ctx.save_for_backward(x2, w32, b32, stats)
        ctx.act, ctx.shape, ctx.wdtype = int(act), shape, weight.dtype
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, b32, stats = ctx.saved_tensors
        n, c = x2.shape
        dy2 = dy.reshape(-1, c).to(x2.dtype).contiguous()
        dx = torch.empty_like(x2)
        dgamma = torch.empty((c,), dtype=torch.float32, device=x2.device)  # overwritten by the kernel
        dbeta = torch.empty((c,), dtype=torch.float32, device=x2.device)
        nbytes = L.lib.ococc_layernorm_act_bwd_workspace_bytes(n, c)
        ws = L.workspace(nbytes, x2.device)
        L.check(L.lib.ococc_layernorm_act_bwd(L.ptr(x2), L.ptr(dy2), n, c, L.ptr(w32), L.ptr(b32),
                                              L.ptr(stats), ctx.act, L.ptr(dx), L.ptr(dgamma),
                                              L.ptr(dbeta), L.dtype_code(x2.dtype), L.ptr(ws),
                                              ws.numel(), L.stream()), 'layernorm_act_bwd')
        return dx.reshape(ctx.shape), dgamma.to(ctx.wdtype), dbeta.to(ctx.wdtype), None, None


def layer_norm_act(x, weight, bias, eps=1e-5, act='none'):
    """y = act(LayerNorm(x)) over the last dim; act in {'none', 'gelu'} (exact erf GELU)."""
    return _LayerNormAct.apply(x, weight, bias, eps, {'none': 0, None: 0, 'gelu': 1}[act])


@NORM_LAYERS.register_module('LN')
class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm over the channel dim with the HIP kernel; ``fused_act='gelu'`` folds
    the following GELU into the same pass (set by the builders below when they see the
    reference's norm -> GELU order)."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, fused_act='none'):
        super().__init__(normalized_shape, eps=eps, elementwise_affine=elementwise_affine)
        assert len(self.normalized_shape) == 1 and elementwise_affine
        self.fused_act = fused_act

    def forward(self, x):
        return layer_norm_act(x, self.weight, self.bias, self.eps, self.fused_act)

    def extra_repr(self):
        return super().extra_repr() + f', fused_act={self.fused_act}'
