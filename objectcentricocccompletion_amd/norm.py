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

from . import _deferred
from . import _lib as L
from .registry import NORM_LAYERS


class _LayerNormAct(Function):

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act, drop_thr=0, seed=0):
        L.require_device(x, weight, bias)
        shape = x.shape
        c = shape[-1]
        x2 = x.reshape(-1, c).contiguous()
        n = x2.size(0)
        dt = L.dtype_code(x2.dtype)
        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        y = torch.empty_like(x2)
        stats = torch.empty((n, 2), dtype=torch.float32, device=x2.device)
        if drop_thr:
            L.check(L.lib.ococc_layernorm_act_dropout_fwd_bf16(L.ptr(x2), n, c, L.ptr(w32), L.ptr(b32), float(eps), int(act),
                                                               int(drop_thr), int(seed), L.ptr(y), L.ptr(stats), L.stream()),
                    'layernorm_act_dropout_fwd')
        else:
            L.check(L.lib.ococc_layernorm_act_fwd(L.ptr(x2), n, c, L.ptr(w32), L.ptr(b32), float(eps),
                                                  int(act), L.ptr(y), L.ptr(stats), dt, L.stream()),
                    'layernorm_act_fwd')
        ctx.save_for_backward(x2, w32, b32, stats, weight, bias)
        ctx.act, ctx.shape, ctx.wdtype, ctx.drop = int(act), shape, weight.dtype, (int(drop_thr), int(seed))
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, b32, stats, weight, bias = ctx.saved_tensors
        n, c = x2.shape
        dy2 = dy.reshape(-1, c).to(x2.dtype).contiguous()
        dx = torch.empty_like(x2)
        dgamma, dbeta = layernorm_act_backward(x2, dy2, w32, b32, stats, ctx.act, dx, weight, bias, drop=ctx.drop)
        if dgamma is None:  # queued: weight.grad / bias.grad receive the sums when the pass ends (_deferred)
            return dx.reshape(ctx.shape), None, None, None, None, None, None
        return dx.reshape(ctx.shape), dgamma.to(ctx.wdtype), dbeta.to(ctx.wdtype), None, None, None, None


def _flush_param_reduce(todo):
    """dgamma/dbeta of the queued layers in one launch per 16; jobs = (partials, rows, c, dgb [2, c])."""
    import ctypes
    for lo in range(0, len(todo), 16):
        ch = todo[lo:lo + 16]
        k = len(ch)
        vp, i32 = ctypes.c_void_p * k, ctypes.c_int32 * k
        L.check(L.lib.ococc_layernorm_param_reduce_multi(
            k, vp(*[j[0].data_ptr() for j in ch]), i32(*[j[1] for j in ch]), i32(*[j[2] for j in ch]),
            vp(*[j[3][0].data_ptr() for j in ch]), vp(*[j[3][1].data_ptr() for j in ch]), L.stream()),
            'layernorm_param_reduce_multi')


_deferred.register('ln', _flush_param_reduce)


def layernorm_act_backward(x2, dy2, w32, b32, stats, act, dx, weight=None, bias=None, drop=(0, 0)):
    """dx into `dx`; -> (dgamma, dbeta) f32 [c].  With the parameters given (by our autograd Functions) and a
    backward pass running that accumulates into their .grad, only the per-block partial sums are computed now, the
    two sums join the pass's end-of-backward launch and reach .grad there (_deferred): (None, None) is returned.
    Otherwise they are reduced right behind the kernel."""
    n, c = x2.shape
    dgb = torch.empty((2, c), dtype=torch.float32, device=x2.device)  # overwritten by the kernel
    dgamma, dbeta = dgb[0], dgb[1]
    dt = L.dtype_code(x2.dtype)
    ws = L.workspace(L.lib.ococc_layernorm_act_bwd_workspace_bytes(n, c), x2.device)
    defer = False
    if n > 0 and weight is not None and bias is not None and weight is not bias \
            and _deferred.deferrable(weight, bias):
        rows = L.lib.ococc_layernorm_act_bwd_partial_rows(n, c, dt)
        defer = rows > 0 and _deferred.defer('ln', (ws, rows, c, dgb), [(weight, dgamma), (bias, dbeta)])
    if drop[0]:   # the forward dropped activations behind the norm: the same mask, regenerated from (threshold, seed)
        L.check(L.lib.ococc_layernorm_act_dropout_bwd_bf16(
            L.ptr(x2), L.ptr(dy2), n, c, L.ptr(w32), L.ptr(b32), L.ptr(stats), int(act), int(drop[0]), int(drop[1]),
            L.ptr(dx), None if defer else L.ptr(dgamma), None if defer else L.ptr(dbeta), L.ptr(ws), ws.numel(),
            L.stream()), 'layernorm_act_dropout_bwd')
        return (None, None) if defer else (dgamma, dbeta)
    L.check(L.lib.ococc_layernorm_act_bwd(L.ptr(x2), L.ptr(dy2), n, c, L.ptr(w32), L.ptr(b32), L.ptr(stats),
                                          int(act), L.ptr(dx), None if defer else L.ptr(dgamma),
                                          None if defer else L.ptr(dbeta), dt, L.ptr(ws), ws.numel(), L.stream()),
            'layernorm_act_bwd')
    return (None, None) if defer else (dgamma, dbeta)


def _fusable_dropout(x, p):
    c = x.shape[-1]
    vec = c % 8 == 0 and 16 <= c <= 512 and ((c // 8) & (c // 8 - 1)) == 0
    return (0.0 < p < 1.0 and x.dtype == torch.bfloat16 and (vec or c in (1024, 1536, 2048))
            and not torch.cuda.is_current_stream_capturing())   # (the seed is a host number: one mask per capture)


def layer_norm_act(x, weight, bias, eps=1e-5, act='none', dropout=0.0):
    """y = dropout(act(LayerNorm(x))) over the last dim; act in {'none', 'gelu'} (exact erf GELU).  ``dropout`` > 0
    (training): inside the kernels for bf16 rows (mask regenerated in the backward from a seed drawn from torch's
    CPU generator), torch's dropout behind the kernel otherwise."""
    code = {'none': 0, None: 0, 'gelu': 1}[act]
    if dropout and _fusable_dropout(x, dropout):
        thr = int(round(dropout * 65536))
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())   # CPU generator: reproducible under manual_seed, no sync
        return _LayerNormAct.apply(x, weight, bias, eps, code, thr, seed)
    y = _LayerNormAct.apply(x, weight, bias, eps, code)
    return torch.nn.functional.dropout(y, dropout, True) if dropout else y


@NORM_LAYERS.register_module('LN')
class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm over the channel dim with the HIP kernel; ``fused_act='gelu'`` folds
    the following GELU into the same pass (set by the builders below when they see the
    reference's norm -> GELU order); ``fused_dropout`` = p of a Dropout that follows the activation (build_mlp sets
    it and leaves a placeholder in the Sequential), applied in training mode only."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, fused_act='none', fused_dropout=0.0):
        super().__init__(normalized_shape, eps=eps, elementwise_affine=elementwise_affine)
        assert len(self.normalized_shape) == 1 and elementwise_affine
        self.fused_act = fused_act
        self.fused_dropout = float(fused_dropout)

    def forward(self, x):
        return layer_norm_act(x, self.weight, self.bias, self.eps, self.fused_act,
                              self.fused_dropout if self.training else 0.0)

    def extra_repr(self):
        extra = f', fused_dropout={self.fused_dropout}' if self.fused_dropout else ''
        return super().extra_repr() + f', fused_act={self.fused_act}' + extra


class FoldedDropout(nn.Module):
    """Placeholder at the position of an nn.Dropout whose work the LayerNorm in front does (keeps the child indices
    of the reference's Sequential(Linear, norm, act, Dropout))."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        return x

    def extra_repr(self):
        return f'p={self.p} (inside the preceding LayerNorm kernel)'


class _AllGatherSum(Function):
    """Sum of a small vector over ranks with a differentiable backward (all_gather forward, all_reduce backward,
    mmdet3d/ops/norm.py:9-24) -- [2C] statistics, the only collective of the SST use_bn option."""

    @staticmethod
    def forward(ctx, input):
        import torch.distributed as dist
        parts = [torch.zeros_like(input) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, input, async_op=False)
        return torch.stack(parts, dim=0).sum(dim=0)

    @staticmethod
    def backward(ctx, grad_output):
        import torch.distributed as dist
        grad_output = grad_output.contiguous()
        dist.all_reduce(grad_output, async_op=False)
        return grad_output


class _NaiveSyncBNMixin(object):
    """Batch statistics averaged over ranks as plain means of per-rank means (mmdet3d/ops/norm.py:28-163);
    single process / eval: the stock BatchNorm."""

    def _sync_forward(self, input, dims, shape):
        import torch.distributed as dist
        assert input.dtype == torch.float32, f'input should be in float32 type, got {input.dtype}'
        assert input.shape[0] > 0, 'SyncBN does not support empty inputs'
        C = input.shape[1]
        mean = torch.mean(input, dim=dims)
        meansqr = torch.mean(input * input, dim=dims)
        vec = _AllGatherSum.apply(torch.cat([mean, meansqr], dim=0)) * (1.0 / dist.get_world_size())
        mean, meansqr = torch.split(vec, C)
        var = meansqr - mean * mean
        self.running_mean += self.momentum * (mean.detach() - self.running_mean)
        self.running_var += self.momentum * (var.detach() - self.running_var)
        invstd = torch.rsqrt(var + self.eps)
        scale = self.weight * invstd
        bias = self.bias - mean * scale
        return input * scale.reshape(shape) + bias.reshape(shape)

    @staticmethod
    def _synced():
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


@NORM_LAYERS.register_module('naiveSyncBN1d')
class NaiveSyncBatchNorm1d(nn.BatchNorm1d, _NaiveSyncBNMixin):

    def forward(self, input):
        if not self._synced() or not self.training:
            return super().forward(input)
        if input.dim() == 2:
            return self._sync_forward(input.unsqueeze(2), [0, 2], (1, -1, 1)).squeeze(2)
        return self._sync_forward(input, [0, 2], (1, -1, 1))


@NORM_LAYERS.register_module('naiveSyncBN2d')
class NaiveSyncBatchNorm2d(nn.BatchNorm2d, _NaiveSyncBNMixin):

    def forward(self, input):
        if not self._synced() or not self.training:
            return super().forward(input)
        return self._sync_forward(input, [0, 2, 3], (1, -1, 1, 1))


@NORM_LAYERS.register_module('naiveSyncBN3d')
class NaiveSyncBatchNorm3d(nn.BatchNorm3d, _NaiveSyncBNMixin):
    """mmdet3d/ops/norm.py:145-198 (5-D tensors)"""

    def forward(self, input):
        if not self._synced() or not self.training:
            return super().forward(input)
        return self._sync_forward(input, [0, 2, 3, 4], (1, -1, 1, 1, 1))


AllReduce = _AllGatherSum   # the reference's name (mmdet3d/ops/norm.py:9-24; imported by models/occ/occ_base.py)
