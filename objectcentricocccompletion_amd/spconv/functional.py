"""Autograd wrappers -- host mirror of mmdet3d/ops/spconv/functional.py:20-101."""
from torch.autograd import Function

from . import ops


class _IndiceConvBase(Function):
    INVERSE = False
    SUBM = False

    @classmethod
    def _fwd(cls, ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        saved = {}
        out = ops.indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out,
                              cls.INVERSE, cls.SUBM, _saved=saved)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters, saved['x_bf16'])
        return out

    @classmethod
    def _bwd(cls, ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters, x_bf16 = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(
            features, filters, grad_output.contiguous(), indice_pairs, indice_pair_num,
            cls.INVERSE, cls.SUBM, _x_bf16=x_bf16, need_input_grad=ctx.needs_input_grad[0],
            need_filter_grad=ctx.needs_input_grad[1], _autograd=True)
        return input_bp, filters_bp, None, None, None


class SparseConvFunction(_IndiceConvBase):

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SparseConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num,
                                       num_activate_out)

    @staticmethod
    def backward(ctx, grad_output):
        return SparseConvFunction._bwd(ctx, grad_output)


class SparseInverseConvFunction(_IndiceConvBase):
    INVERSE = True

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SparseInverseConvFunction._fwd(ctx, features, filters, indice_pairs,
                                              indice_pair_num, num_activate_out)

    @staticmethod
    def backward(ctx, grad_output):
        return SparseInverseConvFunction._bwd(ctx, grad_output)


class SubMConvFunction(_IndiceConvBase):
    SUBM = True

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SubMConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num,
                                     num_activate_out)

    @staticmethod
    def backward(ctx, grad_output):
        return SubMConvFunction._bwd(ctx, grad_output)


class _IndiceConvLN(Function):
    """conv -> LayerNorm -> act with the norm in the conv kernel's epilogue (forward) and the existing LN
    backward kernel in front of indice_conv_backward (backward).  mode: (inverse, subm)."""

    @staticmethod
    def forward(ctx, features, filters, gamma, beta, indice_pairs, indice_pair_num, num_activate_out, eps, act,
                inverse, subm):
        saved = {}
        res = ops.indice_conv_ln(features, filters, gamma, beta, eps, act, indice_pairs, indice_pair_num,
                                 num_activate_out, inverse, subm, _saved=saved)
        if res is None:
            raise ops.L.OcoccError('no fused conv+LN kernel for this shape')
        conv_out, y, stats = res
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters, saved['x_bf16'], conv_out, stats,
                              saved['g32'], saved['b32'], gamma, beta)
        ctx.meta = (int(act), bool(inverse), bool(subm), gamma.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch
        L = ops.L
        indice_pairs, indice_pair_num, features, filters, x_bf16, conv_out, stats, g32, b32, gamma, beta = \
            ctx.saved_tensors
        act, inverse, subm, wdtype = ctx.meta
        n, c = conv_out.shape
        dy2 = dy.to(torch.bfloat16).contiguous()
        dconv = torch.empty_like(conv_out)
        from ..norm import layernorm_act_backward
        dgamma, dbeta = layernorm_act_backward(conv_out, dy2, g32, b32, stats, act, dconv, gamma, beta)
        input_bp, filters_bp = ops.indice_conv_backward(
            features, filters, dconv, indice_pairs, indice_pair_num, inverse, subm, _x_bf16=x_bf16,
            need_input_grad=ctx.needs_input_grad[0], need_filter_grad=ctx.needs_input_grad[1], _autograd=True)
        return (input_bp, filters_bp, None if dgamma is None else dgamma.to(wdtype),
                None if dbeta is None else dbeta.to(wdtype), None, None, None, None, None, None, None)


def indice_conv_ln(features, filters, gamma, beta, indice_pairs, indice_pair_num, num_activate_out, eps, act,
                   inverse=False, subm=False):
    return _IndiceConvLN.apply(features, filters, gamma, beta, indice_pairs, indice_pair_num, num_activate_out,
                               eps, act, inverse, subm)


indice_conv = SparseConvFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
