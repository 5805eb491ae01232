"""TEST INFRASTRUCTURE (oracle) -- not part of the product; only tests/ import it.

CPU restatement, in float64, of the occupancy decoder's per-query MLP:

  pos_encode     PosEncode.forward                 mmdet3d/models/occ/occ_base.py:33-57   (float32, as the reference)
  decoder        OccDecoder.occ_forward            mmdet3d/models/occ/occ_base.py:99-153
                 = [LN(roi feature) | pos_encode(xyz)] through build_mlp's Sequential(Linear(bias=False), LN, GELU
                 [, Dropout]) blocks and the Linear head (mmdet3d/ops/sst/sst_ops.py:333-360), eval mode
  mlp_layer      one block, y = GELU(LN(x W^T + bias + add[idx])), optionally followed by the head's dot product

``rounding``:
  None     nowhere -- the reference's fp32 arithmetic; pinned against tests/golden/ococc_head.npz (dec_logits, generated
           from the imported reference by oracle/gen_golden_ococc.py) in tests/test_decoder_oracle_cpu.py;
  'bf16'   where the fused kernels of csrc/mlp_layer.hip round to bf16: the positional encoding, the Linear weights
           of the per-point GEMMs and every activation a layer hands on (the head reads the rounded activation).  The
           per-RoI half of the first layer (W_roi . LN(f), one row per RoI) stays float32 in the product: not rounded.
Sums (GEMM accumulators, LayerNorm statistics) are float64 here and float32 in the kernels.
"""
import math

import numpy as np
import torch

F64 = torch.float64


def r16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(F64)


def pos_encode(xyz, L=10, bound=(-8.0, -8.0, -4.0, 8.0, 8.0, 4.0), use_norm=True):
    """[M, 3] float32 -> [M, 6 L] float32, the reference's operation order (occ_base.py:39-57)."""
    x = xyz.to(torch.float32)
    if use_norm:
        lo, hi = torch.tensor(bound[:3], dtype=torch.float32), torch.tensor(bound[3:], dtype=torch.float32)
        x = (x - lo) / (hi - lo) * 2.0 - 1.0
    x = x.reshape(-1, 1, 3)
    freq = torch.pow(2, torch.linspace(0.0, L - 1, L))
    x = x * freq.view(1, L, 1)
    x = torch.cat([torch.sin(np.pi * x), torch.cos(np.pi * x)], dim=1)
    return x.reshape(xyz.shape[0], -1)


def layer_norm(z, w, b, eps):
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    return (z - mu) / torch.sqrt(var + eps) * w.to(F64) + b.to(F64)


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mlp_layer(x, W, ln_w, ln_b, eps, bias=None, add=None, idx=None, head_w=None, head_b=None, rounding=None, act='gelu'):
    """x [M, k], W [n, k] -> (y [M, n] float64 (bf16 values under rounding='bf16'), head [M] or None)"""
    rd = r16 if rounding == 'bf16' else (lambda t: t.to(F64))
    z = rd(x) @ rd(W).t()
    if bias is not None:
        z = z + bias.to(F64)
    if add is not None:
        z = z + add.to(F64)[idx.long()]
    y = layer_norm(z, ln_w, ln_b, eps) if ln_w is not None else z
    if act == 'gelu':
        y = gelu(y)
    y = rd(y)
    head = None
    if head_w is not None:
        head = y @ head_w.to(F64).view(-1) + (0.0 if head_b is None else head_b.to(F64).view(()))
    return y, head


def decoder(P, prefix, roi_feats, xyz, idx, L=10, eps=1e-3, use_ln=True, ln_eps=1e-5, rounding=None):
    """Logits [M] of query points xyz [M, 3] of RoIs idx [M]; P: state dict (torch tensors), ``prefix`` the decoder's key
    prefix ('' or '....occ_decoder.').  The first Linear's columns are applied as the reference applies them, on
    cat(LN(f)[idx], pe); under rounding='bf16' its per-RoI part is kept apart (float32 in the product, see above)."""
    g = lambda k: P[prefix + k]
    f = roi_feats.to(F64)
    if use_ln:
        f = layer_norm(f, g('ln.weight'), g('ln.bias'), ln_eps)
    blocks = sorted({int(k[len(prefix):].split('.')[1]) for k in P if k.startswith(prefix + 'conv_occ.')})
    hidden, last = blocks[:-1], blocks[-1]
    pe = pos_encode(xyz, L)
    D = roi_feats.shape[1]
    W0 = g('conv_occ.0.0.weight')
    roi_part = f @ W0[:, :D].to(F64).t()
    if rounding == 'bf16':
        roi_part = roi_part.to(torch.float32).to(F64)
    x, head = pe, None
    for i in hidden:
        W = g(f'conv_occ.{i}.0.weight')
        is_last = i == hidden[-1]
        x, head = mlp_layer(x, W[:, D:] if i == 0 else W, g(f'conv_occ.{i}.1.weight'), g(f'conv_occ.{i}.1.bias'), eps,
                            add=roi_part if i == 0 else None, idx=idx if i == 0 else None,
                            head_w=g(f'conv_occ.{last}.weight') if is_last else None,
                            head_b=g(f'conv_occ.{last}.bias') if is_last else None, rounding=rounding)
    return head
