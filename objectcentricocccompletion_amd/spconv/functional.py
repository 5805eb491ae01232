"""Autograd wrappers -- host mirror of mmdet3d/ops/spconv/functional.py:20-101."""
from torch.autograd import Function

from . import ops


class LnBackwardLink(object):
    """What the dgrad of the NEXT conv layer needs to run this block's LayerNorm backward in its epilogue, and where it
    leaves the result.  Created by ``indice_conv_ln`` when a chain of blocks is declared (``chain_ln_backward``)."""
    __slots__ = ('conv_out', 'stats', 'g32', 'b32', 'act', 'fused', 'partials', 'rows', 'expect')

    def __init__(self):
        self.conv_out = self.stats = self.g32 = self.b32 = self.partials = self.expect = None
        self.act, self.fused, self.rows = 0, False, 0


class chain_ln_backward(object):
    """Inside this context the output of a conv -> LN -> act block (``indice_conv_ln``) that goes STRAIGHT into the next
    block's convolution, and nowhere else, has its LayerNorm backward fused into that convolution's input-gradient
    kernel (ococc_sparse_conv_tile_lnbwd_bf16): the gradient that travels between the two autograd nodes is then the
    gradient of the block's CONV output, not of its activation output -- hooks on the tensor in between would see
    that.  The promise "nowhere else" is the caller's (occ_encoder.SubMOccEncoder: a plain stack); a second consumer
    is detected in the backward pass (the incoming gradient is not the buffer the fused kernel wrote) and raises."""
    active = False

    def __enter__(self):
        self._prev, chain_ln_backward.active = chain_ln_backward.active, True
        return self

    def __exit__(self, *exc):
        chain_ln_backward.active = self._prev
        return False


class _IndiceConvBase(Function):
    INVERSE = False
    SUBM = False

    @classmethod
    def _fwd(cls, ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        saved = {}
        out = ops.indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out,
                              cls.INVERSE, cls.SUBM, _saved=saved)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters, saved['x_bf16'])
        # (a conv -> LN -> act block in front, in a declared chain: see LnBackwardLink below)
        ctx.link_in = getattr(features, '_ococc_ln_link', None) if chain_ln_backward.active else None
        return out

    @classmethod
    def _bwd(cls, ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters, x_bf16 = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(
            features, filters, grad_output.contiguous(), indice_pairs, indice_pair_num,
            cls.INVERSE, cls.SUBM, _x_bf16=x_bf16, need_input_grad=ctx.needs_input_grad[0],
            need_filter_grad=ctx.needs_input_grad[1], _autograd=True, _ln_link=getattr(ctx, 'link_in', None))
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
    backward kernel in front of indice_conv_backward (backward) -- or, in a declared chain, inside the dgrad kernel of
    the next layer (LnBackwardLink).  mode: (inverse, subm)."""

    @staticmethod
    def forward(ctx, features, filters, gamma, beta, indice_pairs, indice_pair_num, num_activate_out, eps, act,
                inverse, subm, link_in, link_out):
        saved = {}
        res = ops.indice_conv_ln(features, filters, gamma, beta, eps, act, indice_pairs, indice_pair_num,
                                 num_activate_out, inverse, subm, _saved=saved)
        if res is None:
            raise ops.L.OcoccError('no fused conv+LN kernel for this shape')
        conv_out, y, stats = res
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters, saved['x_bf16'], conv_out, stats,
                              saved['g32'], saved['b32'], gamma, beta)
        ctx.meta = (int(act), bool(inverse), bool(subm), gamma.dtype)
        ctx.link_in, ctx.link_out = link_in, link_out
        if link_out is not None:
            link_out.conv_out, link_out.stats, link_out.g32, link_out.b32 = conv_out, stats, saved['g32'], saved['b32']
            link_out.act = int(act)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch
        L = ops.L
        indice_pairs, indice_pair_num, features, filters, x_bf16, conv_out, stats, g32, b32, gamma, beta = \
            ctx.saved_tensors
        act, inverse, subm, wdtype = ctx.meta
        n, c = conv_out.shape
        mine = ctx.link_out
        if mine is not None and mine.fused:
            # the next layer's dgrad already ran this block's LN backward: dy IS d conv_out
            if (dy.data_ptr(), dy._version) != mine.expect or dy.dtype != torch.bfloat16:
                from .. import _deferred
                _deferred.discard()   # (this pass ends here: its queued reductions must not run in the next one)
                raise L.OcoccError('chain_ln_backward: the output of a conv -> LN -> act block had a second consumer '
                                   '(its gradient is not the untouched buffer the fused dgrad kernel wrote)')
            dconv = dy
            from .. import _deferred
            dgb = torch.empty((2, c), dtype=torch.float32, device=dy.device)
            if gamma is not beta and _deferred.deferrable(gamma, beta) and \
                    _deferred.defer('ln', (mine.partials, mine.rows, c, dgb), [(gamma, dgb[0]), (beta, dgb[1])]):
                dgamma = dbeta = None
            else:
                from ..norm import _flush_param_reduce
                _flush_param_reduce([(mine.partials, mine.rows, c, dgb)])
                dgamma, dbeta = dgb[0], dgb[1]
            mine.fused, mine.partials, mine.expect = False, None, None
        else:
            dy2 = dy.to(torch.bfloat16).contiguous()
            dconv = torch.empty_like(conv_out)
            from ..norm import layernorm_act_backward
            dgamma, dbeta = layernorm_act_backward(conv_out, dy2, g32, b32, stats, act, dconv, gamma, beta)
        input_bp, filters_bp = ops.indice_conv_backward(
            features, filters, dconv, indice_pairs, indice_pair_num, inverse, subm, _x_bf16=x_bf16,
            need_input_grad=ctx.needs_input_grad[0], need_filter_grad=ctx.needs_input_grad[1], _autograd=True,
            _ln_link=ctx.link_in)
        return (input_bp, filters_bp, None if dgamma is None else dgamma.to(wdtype),
                None if dbeta is None else dbeta.to(wdtype), None, None, None, None, None, None, None, None, None)


def indice_conv_ln(features, filters, gamma, beta, indice_pairs, indice_pair_num, num_activate_out, eps, act,
                   inverse=False, subm=False):
    link_in = getattr(features, '_ococc_ln_link', None) if chain_ln_backward.active else None
    link_out = LnBackwardLink() if (chain_ln_backward.active and torch_is_grad_enabled()) else None
    y = _IndiceConvLN.apply(features, filters, gamma, beta, indice_pairs, indice_pair_num, num_activate_out,
                            eps, act, inverse, subm, link_in, link_out)
    if link_out is not None:
        y._ococc_ln_link = link_out
    return y


def torch_is_grad_enabled():
    import torch
    return torch.is_grad_enabled()


class SparseMaxPoolFunction(Function):
    """functional.py:77-92 of the reference"""

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        out = ops.indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, out = ctx.saved_tensors
        # (the rulebook's device-side tables hang on the indice_pairs OBJECT the forward saw; saved tensors come back as
        # new objects, so the table for the backward direction is derived once more from the pairs)
        return ops.indice_maxpool_backward(features, out, grad_output, indice_pairs, indice_pair_num), None, None, None


indice_conv = SparseConvFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_maxpool = SparseMaxPoolFunction.apply
