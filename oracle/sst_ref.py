"""TEST INFRASTRUCTURE (oracle) -- not part of the product; only tests/ import it.

CPU restatement, in float64, of one SST encoder layer and of the window partition / positional embedding around it:

  window_ids        get_window_coors                       mmdet3d/ops/sst/sst_ops.py:266-313
  pos_embed         SSTInputLayerV2.get_pos_embed          mmdet3d/models/middle_encoders/sst_input_layer_v2.py:239-289
  encoder_layer     WindowAttention.forward + EncoderLayer.forward (post-norm)
                                                           mmdet3d/models/sst/sst_basic_block_v2.py:41-75,105-127
                    around torch.nn.MultiheadAttention (q = k = x + pos, v = x, key padding = the window's population)
  encoder_layer_backward   the chain rule of the same layer, written out
  window_attention_core    the attention core alone on padded windows, forward and backward (rounding=None | 'window':
                    the store points of the per-window kernels, csrc/window_attn.hip)

``rounding`` selects where values are rounded to bf16, i.e. which product path is mirrored:
  None     nowhere: the reference's own fp32 arithmetic (pinned against tests/golden/sst.npz, generated from the
           imported reference by oracle/gen_golden_sst.py, at 1e-5);
  'core'   q, k, v, P and the attention output: the reference-shaped f32 block of the product, whose attention core
           (csrc/window_attn.hip) works on bf16 operands;
  'bf16'   every value the fused kernels of csrc/window_block.hip store: x + pos, the weights, q, k, v, P, the attention
           output, y1, act(h), y2, and in the backward every gradient that is an MFMA operand or leaves a kernel;
  'ops'    the bf16 layer run operator by operator (layer_cfg compute_dtype=bf16 on a layer the fused kernels do not
           cover, e.g. the cosine variant): every operator's output is a bf16 tensor -- the projections after a
           bf16-rounded bias, the residual sums, both LayerNorm outputs, the hidden activation before and after the
           non-linearity, the second linear's output.
``cosine=(tau, tau_min)`` selects scaled cosine attention (CosineMultiheadAttention, mmdet3d/models/sst/cosine_msa.py:
123-185,449-466): per-head unit vectors, logits cos / clamp(tau, tau_min); tau a scalar or one value per head.  The product
hands the attention core q^ sqrt(d) / tau and k^ (its fixed 1 / sqrt(d) scale leaves cos / tau), rounded to bf16 where the
core reads bf16 ('core', 'ops').
Sums (GEMM accumulators, softmax, LayerNorm statistics, residual adds) are float64 here and float32 in the kernels.
"""
import math

import numpy as np
import torch

F64 = torch.float64


def window_ids(coors, sparse_shape, window_shape, do_shift):
    """coors [N, 4] (b, z, y, x) int64 -> (batch_win_inds [N], coors_in_win [N, 3] (z, y, x))"""
    wx, wy, wz = window_shape
    sx, sy, sz = sparse_shape
    nx, ny, nz = (int(np.ceil(s / w) + 1) for s, w in ((sx, wx), (sy, wy), (sz, wz)))
    hx, hy, hz = (wx // 2, wy // 2, wz // 2) if do_shift else (wx, wy, wz)
    if sz == wz:
        hz = 0
    x, y, z = coors[:, 3] + hx, coors[:, 2] + hy, coors[:, 1] + hz
    win = coors[:, 0] * (nx * ny * nz) + (x // wx) * (ny * nz) + (y // wy) * nz + z // wz
    return win, torch.stack([z % wz, y % wy, x % wx], -1)


def pos_embed(coors_in_win, window_shape, feat_dim, temperature=10000):
    """[N, 3] (z, y, x) -> [N, feat_dim] float32, as the reference computes it (float32 throughout)."""
    wx, wy, wz = window_shape
    z, y, x = (coors_in_win[:, i].float() - w / 2 for i, w in ((0, wz), (1, wy), (2, wx)))
    n = feat_dim // 3
    inv = torch.arange(n, dtype=torch.float32)
    inv = temperature ** (2 * (inv // 2) / n)
    parts = []
    for v in (x, y, z):
        e = v[:, None] / inv[None, :]
        parts.append(torch.stack([e[:, ::2].sin(), e[:, 1::2].cos()], -1).flatten(1))
    out = torch.cat(parts, -1)
    return torch.cat([out, torch.zeros(out.shape[0], feat_dim - out.shape[1])], 1)


def r16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(F64)


def _gelu(h):
    return 0.5 * h * (1.0 + torch.erf(h / math.sqrt(2.0)))


def _gelu_grad(h):
    return 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0))) + h * torch.exp(-0.5 * h * h) / math.sqrt(2.0 * math.pi)


def _act(name):
    if name == 'gelu':
        return _gelu, _gelu_grad
    return (lambda h: h.clamp(min=0)), (lambda h: (h > 0).to(F64))


def _windows(win):
    """padded window table: idx [nW, T] token index or -1 (tokens of a window in their flat order)"""
    order = torch.argsort(win, stable=True)
    uniq, counts = torch.unique_consecutive(win[order], return_counts=True)
    T = int(counts.max())
    start = torch.cumsum(counts, 0) - counts
    idx = torch.full((len(uniq), T), -1, dtype=torch.long)
    rank = torch.arange(len(order)) - torch.repeat_interleave(start, counts)
    idx[torch.repeat_interleave(torch.arange(len(uniq)), counts), rank] = order
    return idx


def _ln(z, g, b, eps):
    mean = z.mean(-1, keepdim=True)
    var = ((z - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    xh = (z - mean) * rstd
    return xh * g + b, xh, rstd


def _ln_bwd(dy, xh, g, rstd):
    dg = dy * g
    return (dg - dg.mean(-1, keepdim=True) - xh * (dg * xh).mean(-1, keepdim=True)) * rstd


def _params(P, rounding, detach=True):
    f = lambda k: (P[k].detach() if detach else P[k]).to(F64)
    w = (lambda k: r16(f(k))) if rounding == 'bf16' else f
    a = 'win_attn.self_attn.'
    return dict(wqkv=w(a + 'in_proj_weight'), bqkv=f(a + 'in_proj_bias'), wo=w(a + 'out_proj.weight'),
                bo=f(a + 'out_proj.bias'), w1=w('linear1.weight'), b1=f('linear1.bias'), w2=w('linear2.weight'),
                b2=f('linear2.bias'), g1=f('norm1.weight'), be1=f('norm1.bias'), g2=f('norm2.weight'), be2=f('norm2.bias'))


def encoder_layer(x, pos, win, P, num_heads=8, rounding=None, act='gelu', eps=1e-5, keep=False, detach=True, ops_rows=None,
                  cosine=None):
    """x, pos [V, E]; win [V] window id of every token -> y2 [V, E] float64 (and the intermediates when keep).
    ``ops_rows`` (bool [V], with rounding='bf16'): rows whose attention block runs operator by operator in the product
    (windows of more than 64 tokens: bf16 library GEMMs with bf16 biases, the per-window attention kernels, a bf16
    residual sum, a stand-alone LayerNorm) -- the forward of those rows gets that path's store points: projections
    rounded after a bf16-rounded bias, the out-projection and the residual sum rounded before the norm."""
    ops = rounding == 'ops'
    if ops:   # the whole layer operator by operator: the 'bf16' store points + those of ops_rows on every row + the FFN's
        rounding = 'bf16'
        ops_rows = torch.ones(x.shape[0], dtype=torch.bool)
    rq = r16 if rounding in ('core', 'bf16') else (lambda t: t)      # operands of the attention core
    ra = r16 if rounding == 'bf16' else (lambda t: t)                # everything else the fused kernels store
    ro = r16 if ops else (lambda t: t)                               # operator outputs of the bf16 operator path
    p = _params(P, rounding, detach)   # detach=False: the parameters stay on the autograd tape (tests)
    x, pos = x.to(F64), pos.to(F64)
    if ops:
        x, pos = r16(x), r16(pos)
    V, E = x.shape
    D = E // num_heads
    xp = ra(x + pos)
    # (cosine attention in the f32 block: q and k are normalised in f32 and only then rounded for the core)
    rqk = (lambda t: t) if (cosine is not None and not ops) else rq
    q = rqk(xp @ p['wqkv'][:E].t() + p['bqkv'][:E])
    k = rqk(xp @ p['wqkv'][E:2 * E].t() + p['bqkv'][E:2 * E])
    v = rq(x @ p['wqkv'][2 * E:].t() + p['bqkv'][2 * E:])
    if ops_rows is not None:
        sel = ops_rows[:, None]
        b16 = r16(p['bqkv'])
        q = torch.where(sel, r16(xp @ p['wqkv'][:E].t() + b16[:E]), q)
        k = torch.where(sel, r16(xp @ p['wqkv'][E:2 * E].t() + b16[E:2 * E]), k)
        v = torch.where(sel, r16(x @ p['wqkv'][2 * E:].t() + b16[2 * E:]), v)
    if cosine is not None:
        tau, tau_min = cosine
        tau = torch.as_tensor(tau, dtype=F64).reshape(-1).clamp(min=tau_min)          # [1] or [H]
        unit = lambda t: t.view(V, num_heads, D) / t.view(V, num_heads, D).norm(dim=-1, keepdim=True).clamp(min=1e-12)
        q = rq((unit(q) * (float(D) ** 0.5 / tau.view(1, -1, 1))).reshape(V, E))
        k = rq(unit(k).reshape(V, E))
    idx = _windows(win)
    nW, T = idx.shape
    valid = idx >= 0
    gi = idx.clamp(min=0)
    qw, kw, vw = (t[gi].view(nW, T, num_heads, D) for t in (q, k, v))
    s = torch.einsum('wthd,wshd->whts', qw, kw) * (float(D) ** -0.5)
    s = s.masked_fill(~valid[:, None, None, :], float('-inf'))
    prob = torch.softmax(s, -1)
    ow = torch.einsum('whts,wshd->wthd', rq(prob), vw).reshape(nW, T, E)
    o = torch.zeros(V, E, dtype=F64)
    o[idx[valid]] = ow[valid]
    o = rq(o)
    z1 = o @ p['wo'].t() + p['bo'] + x
    if ops_rows is not None:
        z1 = torch.where(ops_rows[:, None], r16(r16(o @ p['wo'].t() + r16(p['bo'])) + x), z1)
    y1f, xh1, rstd1 = _ln(z1, p['g1'], p['be1'], eps)
    y1 = ra(y1f)
    fa, fg = _act(act)
    h = ro(y1 @ p['w1'].t() + (r16(p['b1']) if ops else p['b1']))
    a = ra(fa(h))
    z2 = ro(ro(a @ p['w2'].t() + (r16(p['b2']) if ops else p['b2'])) + y1)
    y2f, xh2, rstd2 = _ln(z2, p['g2'], p['be2'], eps)
    y2 = ra(y2f)
    if not keep:
        return y2
    return y2, dict(p=p, x=x, xp=xp, q=q, k=k, v=v, idx=idx, valid=valid, prob=prob, o=o, xh1=xh1, rstd1=rstd1, y1=y1,
                    h=h, a=a, xh2=xh2, rstd2=rstd2, fg=fg, num_heads=num_heads, rounding=rounding, ops_rows=ops_rows)


def encoder_layer_backward(dy2, c):
    """Chain rule of encoder_layer in the fused kernels' order and with their bf16 roundings (rounding='bf16' forward).
    dy2 [V, E] -> dict(dx, and the parameter gradients under the module's parameter names)."""
    p, H = c['p'], c['num_heads']
    r16 = globals()['r16'] if c['rounding'] == 'bf16' else (lambda t: t)   # without roundings: the plain chain rule
    V, E = c['x'].shape
    D = E // H
    dy2 = dy2.to(F64)
    # --- FFN block (token_ffn_block_bwd_kernel)
    dz2 = _ln_bwd(dy2, c['xh2'], p['g2'], c['rstd2'])
    g_n2, b_n2 = (dy2 * c['xh2']).sum(0), dy2.sum(0)
    dz2r = r16(dz2)
    dh = r16((dz2r @ p['w2']) * r16(c['fg'](c['h'])))
    dy1 = r16(dh @ p['w1'] + dz2)
    g_w2, g_b2 = dz2r.t() @ c['a'], dz2r.sum(0)
    g_w1, g_b1 = dh.t() @ c['y1'], dh.sum(0)
    # --- attention block (window_attn_block_bwd_kernel)
    dz1 = _ln_bwd(dy1, c['xh1'], p['g1'], c['rstd1'])
    g_n1, b_n1 = (dy1 * c['xh1']).sum(0), dy1.sum(0)
    dz1r = r16(dz1)
    do = r16(dz1r @ p['wo'])
    g_wo, g_bo = dz1r.t() @ c['o'], dz1r.sum(0)
    idx, valid = c['idx'], c['valid']
    nW, T = idx.shape
    gi = idx.clamp(min=0)
    qw, kw, vw, dow = (t[gi].view(nW, T, H, D) for t in (c['q'], c['k'], c['v'], do))
    dow = dow * valid[:, :, None, None]
    prob = c['prob']
    dp = torch.einsum('wthd,wshd->whts', dow, vw)
    delta = (prob * dp).sum(-1, keepdim=True)
    if c.get('ops_rows') is not None:
        # rows of the operator path: their attention core is csrc/window_attn.hip, whose backward takes
        # delta = sum(dO * O) from the STORED (bf16) output (window_attention_core(rounding='window') below)
        ow = c['o'][gi].view(nW, T, H, D)
        delta_o = (dow * ow).sum(-1).permute(0, 2, 1)[..., None]                     # [nW, H, T, 1]
        delta = torch.where((c['ops_rows'][gi] & valid)[:, None, :, None], delta_o, delta)
    ds = r16(prob * (dp - delta) * (float(D) ** -0.5))
    ds = ds * valid[:, None, :, None]
    dq = r16(torch.einsum('whts,wshd->wthd', ds, kw)).reshape(nW, T, E)
    dk = r16(torch.einsum('whts,wthd->wshd', ds, qw)).reshape(nW, T, E)
    pq = r16(prob) * valid[:, None, :, None]
    dv = r16(torch.einsum('whts,wthd->wshd', pq, dow)).reshape(nW, T, E)
    dqkv = torch.zeros(V, 3 * E, dtype=F64)
    dqkv[idx[valid]] = torch.cat([dq, dk, dv], -1)[valid]
    dx = r16(dqkv @ p['wqkv'] + dz1)
    g_wqkv = torch.cat([dqkv[:, :2 * E].t() @ c['xp'], dqkv[:, 2 * E:].t() @ c['x']], 0)
    a = 'win_attn.self_attn.'
    return {'dx': dx, a + 'in_proj_weight': g_wqkv, a + 'in_proj_bias': dqkv.sum(0), a + 'out_proj.weight': g_wo,
            a + 'out_proj.bias': g_bo, 'linear1.weight': g_w1, 'linear1.bias': g_b1, 'linear2.weight': g_w2,
            'linear2.bias': g_b2, 'norm1.weight': g_n1, 'norm1.bias': b_n1, 'norm2.weight': g_n2, 'norm2.bias': b_n2}


def sst_blocks(feats, coors, sd, sparse_shape, window_shape, num_blocks=2, num_heads=8, rounding=None, act='gelu',
               ops_windows_above=None):
    """SSTv2 without the attached convolutions (to_bev=False): blocks of two encoder layers, the second of each on the
    shifted windows (BasicShiftBlockV2, sst_basic_block_v2.py:130-169; SSTv2.forward, backbones/sst_v2.py:115-154).
    sd: state dict with the module's names (block_list.{i}.encoder_list.{j}....).  No voxel may be dropped."""
    out = feats.to(F64)
    wins = [window_ids(coors, sparse_shape, window_shape, i == 1) for i in range(2)]
    poss = [pos_embed(w[1], window_shape, feats.shape[1]).to(F64) for w in wins]
    if rounding == 'bf16':
        out, poss = r16(out), [r16(t) for t in poss]
    # ops_windows_above (with rounding='bf16'): the rows of windows with more tokens than this run operator by operator in
    # the product (64: the fused kernels' tile) -- see encoder_layer(ops_rows=)
    big = [None, None]
    if ops_windows_above is not None:
        for j in range(2):
            _, inv, cnt = torch.unique(wins[j][0], return_inverse=True, return_counts=True)
            big[j] = cnt[inv] > ops_windows_above
    for b in range(num_blocks):
        for j in range(2):
            pre = f'block_list.{b}.encoder_list.{j}.'
            P = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
            out = encoder_layer(out, poss[j], wins[j][0], P, num_heads, rounding, act, ops_rows=big[j])
    return out


def window_attention_core(q, k, v, key_len, num_heads, dout=None, rounding=None):
    """softmax(q k^T / sqrt(d) + key mask) v on padded windows [nW, T, C] in float64, and -- with ``dout`` -- its
    backward (dq, dk, dv).  Follows mmdet3d/models/backbones/sst_basic_block_v2.py:41-75 (nn.MultiheadAttention's core
    on the padded window layout; key_padding_mask as a length).
    rounding='window' mirrors the store points of csrc/window_attn.hip (the per-window kernels: windows of more than 64
    tokens, the cosine variant): the normalised probabilities are rounded to bf16 before P V, the output is stored in
    bf16; in the backward delta = sum(dO * O) uses the bf16 output, dS = P (dP - delta) scale is rounded to bf16 before
    the dQ / dK products, P before the dV product, and dq, dk, dv are stored in bf16.  q, k, v, dout are expected to hold
    bf16-representable values (the kernels read bf16)."""
    rw = r16 if rounding == 'window' else (lambda t: t)
    nW, T, C = q.shape
    H = num_heads
    D = C // H
    scale = float(D) ** -0.5
    q4, k4, v4 = (t.double().view(nW, T, H, D) for t in (q, k, v))
    mask = torch.arange(T, device=q.device)[None, :] >= key_len.to(q.device)[:, None].long()
    s = torch.einsum('wthd,wshd->whts', q4, k4) * scale
    s = s.masked_fill(mask[:, None, None, :], float('-inf'))
    p = torch.softmax(s, -1)
    p = torch.where(torch.isnan(p), torch.zeros_like(p), p)        # (a window without keys: zeros, as the kernel)
    o = rw(torch.einsum('whts,wshd->wthd', rw(p), v4)).reshape(nW, T, C)
    if dout is None:
        return o
    do4 = dout.double().view(nW, T, H, D)
    delta = (do4 * o.view(nW, T, H, D)).sum(-1)                    # [nW, T, H]
    dp = torch.einsum('wthd,wshd->whts', do4, v4)
    ds = rw(p * (dp - delta.permute(0, 2, 1)[..., None]) * scale)
    dq = rw(torch.einsum('whts,wshd->wthd', ds, k4)).reshape(nW, T, C)
    dk = rw(torch.einsum('whts,wthd->wshd', ds, q4)).reshape(nW, T, C)
    dv = rw(torch.einsum('whts,wthd->wshd', rw(p), do4)).reshape(nW, T, C)
    return o, dq, dk, dv
