"""Temporal transformer pieces -- host mirror of mmdet3d/models/occ/layers.py:
PositionalEncoding (:8-32), SimpleEncoderLayer (:35-87), TransformerEncoder (:89-99).  (The reference file also holds
a TransformerDecoder / SimpleDecoderLayer that nothing in ococcnet.py instantiates: not on the path, not built.)
Parameter names match nn.MultiheadAttention / the reference (self_attn.in_proj_weight, ...,
linear1, linear2, norm1, norm2) so checkpoints load."""
import copy
import math

import torch
from torch import nn

from ..norm import layer_norm_act
from ..sst.sst_ops import get_activation_layer


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
    Supports the boolean attn_mask / key_padding_mask forms the reference passes."""

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

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None):
        L, B, E = query.shape
        S = key.shape[0]
        H, D = self.num_heads, self.head_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        q = torch.addmm(b[:E], query.reshape(L * B, E), w[:E].t())
        k = torch.addmm(b[E:2 * E], key.reshape(S * B, E), w[E:2 * E].t())
        v = torch.addmm(b[2 * E:], value.reshape(S * B, E), w[2 * E:].t())
        q = q.view(L, B * H, D).transpose(0, 1) * (D ** -0.5)
        k = k.view(S, B * H, D).transpose(0, 1)
        v = v.view(S, B * H, D).transpose(0, 1)
        att = torch.bmm(q, k.transpose(1, 2))  # [B*H, L, S]
        if attn_mask is not None:
            att = att.masked_fill(attn_mask[None], float('-inf')) if attn_mask.dtype == torch.bool \
                else att + attn_mask[None]
        if key_padding_mask is not None:
            att = att.view(B, H, L, S).masked_fill(key_padding_mask[:, None, None, :], float('-inf'))
            att = att.view(B * H, L, S)
        att = torch.softmax(att, dim=-1)
        if self.dropout > 0 and self.training:
            att = torch.nn.functional.dropout(att, self.dropout)
        out = torch.bmm(att, v).transpose(0, 1).reshape(L * B, E)
        out = self.out_proj(out).view(L, B, E)
        return out, None


class SimpleEncoderLayer(nn.Module):
    """Post-LN encoder layer with q = k = src + pos, v = src (layers.py:35-87)."""

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

    def _norm(self, norm, x):  # nn.LayerNorm parameters, HIP kernel
        return layer_norm_act(x, norm.weight, norm.bias, norm.eps, 'none')

    def forward(self, src, key_padding_mask=None, pos_enc=None, attn_mask=None):
        q = k = self.with_pos_embed(src, pos_enc)
        src2 = self.self_attn(q, k, value=src, attn_mask=attn_mask, key_padding_mask=key_padding_mask)[0]
        src = self._norm(self.norm1, src + self.dropout1(src2))
        src2 = self.linear2(self.dropout(self.activation(self.linear1(src))))
        src = self._norm(self.norm2, src + self.dropout2(src2))
        return src


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    def forward(self, src, key_padding_mask=None, pos_enc=None, attn_mask=None):
        output = src
        for layer in self.layers:
            output = layer(output, key_padding_mask, pos_enc, attn_mask)
        return output
