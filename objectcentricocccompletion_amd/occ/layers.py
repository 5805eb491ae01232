"""Temporal transformer pieces -- host mirror of mmdet3d/models/occ/layers.py:
PositionalEncoding (:8-32), SimpleEncoderLayer (:35-87), TransformerEncoder (:89-99).  (The reference file also holds
a TransformerDecoder / SimpleDecoderLayer that nothing in ococcnet.py instantiates: not on the path, not built.)
Parameter names match nn.MultiheadAttention / the reference (self_attn.in_proj_weight, ...,
linear1, linear2, norm1, norm2) so checkpoints load."""
import copy
import math

import torch
from torch import nn

import os

from .. import gemm
from ..norm import layer_norm_act
from ..sst.sst_ops import get_activation_layer

# the attention core (scores, masks, softmax, dropout, @ v) as ONE launch per direction: csrc/causal_attn.hip (round 6).
# 0: the operator chain below (bmm, masked_fill, softmax, dropout, bmm)
FUSED_ATTENTION = os.environ.get('OCOCC_FUSED_ATTENTION', '1') == '1'


class _TemporalAttention(torch.autograd.Function):
    """ctx rows [L B, E] = softmax(mask(q k^T / sqrt(D))) v per (tracklet, head): ococc_temporal_attention_{fwd,bwd}_f32.
    q / k / v: f32 token-major [L B, H D] (q and k may be column slices of one projection); masks: uint8 or None;
    seed: a device int64 tensor (dropout; drawn by torch's generator, so a captured graph draws a new one per replay)."""

    @staticmethod
    def forward(ctx, q, k, v, attn_mask, key_pad, dims, p_drop, seed):
        # ``k`` None: ``q`` is the packed projection [L B, 2 H D] = q | k (the encoder layers: one product for both) -- the
        # gradient then leaves as ONE [L B, 2 H D] tensor too, and no slice backward (a zero fill + a copy per half) runs
        from .. import _lib as L
        B, H, Lq, S, D = dims
        ctx.packed = k is None
        if ctx.packed:
            qk = q
            q, k = qk[:, :H * D], qk[:, H * D:]
        out = torch.empty((Lq * B, H * D), dtype=torch.float32, device=q.device)
        probs = torch.empty((B * H, Lq, S), dtype=torch.float32, device=q.device)
        L.check(L.lib.ococc_temporal_attention_fwd_f32(
            q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), L.ptr(attn_mask), L.ptr(key_pad),
            B, H, Lq, S, D, float(D) ** -0.5, float(p_drop), 0, L.ptr(seed), probs.data_ptr(), out.data_ptr(), out.stride(0),
            L.stream()), 'temporal_attention_fwd')
        ctx.save_for_backward(qk if ctx.packed else q, v if ctx.packed else k, v, probs, out, *(() if seed is None else (seed,)))
        ctx.meta = (dims, float(p_drop), seed is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        from .. import _lib as L
        q, k, v, probs, out = ctx.saved_tensors[:5]
        dims, p_drop, has_seed = ctx.meta
        seed = ctx.saved_tensors[5] if has_seed else None
        B, H, Lq, S, D = dims
        d_out = d_out.contiguous()
        dqk = None
        if ctx.packed:
            qk = q
            q, k = qk[:, :H * D], qk[:, H * D:]
            dqk = torch.empty_like(qk)
            dq, dk = dqk[:, :H * D], dqk[:, H * D:]
        else:
            dq = torch.empty((Lq * B, H * D), dtype=torch.float32, device=q.device)
            dk = torch.empty((S * B, H * D), dtype=torch.float32, device=q.device)
        dv = torch.empty((S * B, H * D), dtype=torch.float32, device=q.device)
        L.check(L.lib.ococc_temporal_attention_bwd_f32(
            q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), B, H, Lq, S, D, float(D) ** -0.5,
            p_drop, 0, L.ptr(seed), probs.data_ptr(), out.data_ptr(), out.stride(0), d_out.data_ptr(), d_out.stride(0),
            dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0), L.stream()),
            'temporal_attention_bwd')
        if ctx.packed:
            return dqk, None, dv, None, None, None, None, None
        return dq, dk, dv, None, None, None, None, None


class PositionalEncoding(nn.Module):
    def __init__(self, d_model: int, max_len: int = 200):
        super().__init__()
        self.d_model = d_model
        self.max_len = max_len

    def forward(self, abs_pos):
        """abs_pos [seq_len, batch] -> [seq_len, batch, d_model] = [sin(t w_i) | cos(t w_i)]."""
        div_term = torch.exp(torch.arange(0, self.d_model, 2, device=abs_pos.device)
                             * (-math.log(10000.0) / self.d_model))
        ang = abs_pos[..., None] * div_term
        return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)


class MultiheadAttention(nn.Module):
    """Self-attention with the parameter layout of nn.MultiheadAttention (in_proj_weight
    [3E,E], in_proj_bias, out_proj.{weight,bias}); sequence-first tensors [L, B, E].
    Supports the boolean attn_mask / key_padding_mask forms the reference passes.

    Own formulation: q and k share their input (x + pos), so they leave ONE product with the first 2E rows of in_proj
    (the reference's nn.MultiheadAttention runs three), heads are taken as views of that [L B, 2E] result; all products go
    through ``gemm`` (f32 as the reference, or bf16 operands on the matrix cores when gemm.GEMM_DTYPE says so)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def _fused_ok(self, q, k, v, L, S, attn_mask, key_padding_mask):
        """the one-launch attention core takes f32 device tensors with 16-byte aligned rows, boolean masks of the shapes
        the reference passes, sequences up to 256 frames (the reference's PositionalEncoding stops at 200), heads up to 384 wide"""
        ok = lambda t: t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
        return (FUSED_ATTENTION and ok(q) and ok(k) and ok(v) and L <= 256 and S <= 256 and self.head_dim % 4 == 0
                and self.head_dim <= 384 and (attn_mask is None or (attn_mask.dtype == torch.bool and attn_mask.shape == (L, S)))
                and (key_padding_mask is None or (key_padding_mask.dtype == torch.bool and key_padding_mask.dim() == 2
                                                  and key_padding_mask.shape[1] == S)))

    def _heads(self, t, n):
        """[n * B, E] token-major -> [B * H, n, D]"""
        return t.reshape(n, -1, self.head_dim).transpose(0, 1)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None):
        L, B, E = query.shape
        S = key.shape[0]
        H = self.num_heads
        w, b = self.in_proj_weight, self.in_proj_bias
        if key is query:      # (the encoder layers: q = k = src + pos)
            qk = gemm.linear(query.reshape(L * B, E), w[:2 * E], b[:2 * E])
            q, k = qk[:, :E], qk[:, E:]
        else:
            q = gemm.linear(query.reshape(L * B, E), w[:E], b[:E])
            k = gemm.linear(key.reshape(S * B, E), w[E:2 * E], b[E:2 * E])
        v = gemm.linear(value.reshape(S * B, E), w[2 * E:], b[2 * E:])
        p_drop = self.dropout if self.training else 0.0
        if self._fused_ok(q, k, v, L, S, attn_mask, key_padding_mask):
            am = None if attn_mask is None else attn_mask.contiguous().view(torch.uint8)
            kp = None if key_padding_mask is None else key_padding_mask.contiguous().view(torch.uint8)
            seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=q.device) if p_drop > 0 else None
            packed = key is query and qk.is_contiguous()
            ctx = _TemporalAttention.apply(qk if packed else q, None if packed else k, v, am, kp, (B, H, L, S, self.head_dim),
                                           p_drop, seed)
            return gemm.linear(ctx, self.out_proj.weight, self.out_proj.bias).view(L, B, E), None
        scores = gemm.bmm(self._heads(q * (self.head_dim ** -0.5), L), self._heads(k, S).transpose(1, 2))   # [B H, L, S]
        if attn_mask is not None:
            scores = scores.masked_fill(attn_mask[None], float('-inf')) if attn_mask.dtype == torch.bool \
                else scores + attn_mask[None]
        if key_padding_mask is not None:
            scores = scores.view(B, H, L, S).masked_fill(key_padding_mask[:, None, None, :], float('-inf')).view(B * H, L, S)
        prob = torch.softmax(scores, dim=-1)
        if self.dropout > 0 and self.training:
            prob = torch.nn.functional.dropout(prob, self.dropout)
        ctx = gemm.bmm(prob, self._heads(v, S)).transpose(0, 1).reshape(L * B, E)
        return gemm.linear(ctx, self.out_proj.weight, self.out_proj.bias).view(L, B, E), None


class SimpleEncoderLayer(nn.Module):
    """Post-LN encoder layer with q = k = src + pos, v = src (layers.py:35-87): attention block, then feed-forward block,
    each `x <- LayerNorm(x + dropout(block(x)))` with the LayerNorm on the HIP kernel."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='gelu', mlp_dropout=0):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(mlp_dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(mlp_dropout)
        self.dropout2 = nn.Dropout(mlp_dropout)
        self.activation = get_activation_layer(activation)
        self.fp16_enabled = False

    def with_pos_embed(self, tensor, pos):
        return tensor if pos is None else tensor + pos

    def _residual_norm(self, norm, x, branch, drop):
        return layer_norm_act(x + drop(branch), norm.weight, norm.bias, norm.eps, 'none')   # nn.LayerNorm parameters, HIP kernel

    def _feed_forward(self, x):
        hidden = self.dropout(self.activation(gemm.linear(x, self.linear1.weight, self.linear1.bias)))
        return gemm.linear(hidden, self.linear2.weight, self.linear2.bias)

    def forward(self, src, key_padding_mask=None, pos_enc=None, attn_mask=None):
        qk_in = self.with_pos_embed(src, pos_enc)
        attended, _ = self.self_attn(qk_in, qk_in, value=src, attn_mask=attn_mask, key_padding_mask=key_padding_mask)
        x = self._residual_norm(self.norm1, src, attended, self.dropout1)
        return self._residual_norm(self.norm2, x, self._feed_forward(x), self.dropout2)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerEncoder(nn.Module):
    """``num_layers`` copies of an encoder layer applied in turn (layers.py:89-99; parameter names layers.<i>....)"""

    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    def forward(self, src, key_padding_mask=None, pos_enc=None, attn_mask=None):
        x = src
        for layer in self.layers:
            x = layer(x, key_padding_mask=key_padding_mask, pos_enc=pos_enc, attn_mask=attn_mask)
        return x
