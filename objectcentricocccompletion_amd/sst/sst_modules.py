"""SST (single-stride sparse transformer) input layer, window attention blocks and backbone --
host mirror of SSTInputLayerV2 (mmdet3d/models/middle_encoders/sst_input_layer_v2.py:41-330),
WindowAttention / EncoderLayer / BasicShiftBlockV2 (mmdet3d/models/sst/sst_basic_block_v2.py:14-169)
and SSTv2 (mmdet3d/models/backbones/sst_v2.py:17-197).  Same constructor arguments, dictionaries
and parameter names (win_attn.self_attn.in_proj_weight, linear1, norm1, ...).

Device work: window ranks ococc_group_rank_i32; attention core ococc_window_attn_{fwd,bwd}_bf16
(MFMA QK^T / PV on padded windows); projections / FFN are GEMMs through torch."""
import torch
from torch import nn

from .. import _lib as L
from ..norm import layer_norm_act
from ..registry import BACKBONES, MIDDLE_ENCODERS, build_conv_layer, build_norm_layer
from .sst_ops import (LazyWindowDict, flat2window_v2, get_flat2win_inds_v2, get_inner_win_inds, get_window_coors,
                      get_window_coors_both, group_rank,
                      window2flat_v2)


@MIDDLE_ENCODERS.register_module()
class SSTInputLayerV2(nn.Module):
    """Regional grouping, voxel drop / region batching, flat<->window index maps, positional
    embedding and key masks for the two window shifts."""

    def __init__(self, drop_info, window_shape, sparse_shape, shuffle_voxels=True, debug=True,
                 normalize_pos=False, pos_temperature=10000, mute=False):
        super().__init__()
        self.meta_drop_info = drop_info
        self.sparse_shape, self.window_shape = sparse_shape, window_shape
        self.shuffle_voxels, self.debug = shuffle_voxels, debug
        self.normalize_pos, self.pos_temperature, self.mute = normalize_pos, pos_temperature, mute

    def set_drop_info(self):
        meta = self.meta_drop_info
        self.drop_info = (meta[0] if self.training else meta[1]) if isinstance(meta, tuple) else meta

    def _window_id_bound(self, voxel_coors, batch_size):
        """an upper bound of every window id of get_window_coors (the group-rank kernel sizes its key table by it): with
        the batch size given it is host arithmetic, without it ONE read-back of the largest batch index for both shifts
        (it was one per group-rank call: four)"""
        import math
        ws, ss = self.window_shape, self.sparse_shape
        wz = ws[2] if len(ws) == 3 else ss[-1]
        per_batch = 1
        for s_, w_ in ((ss[0], ws[0]), (ss[1], ws[1]), (ss[2], wz)):
            per_batch *= int(math.ceil(s_ / w_) + 1)
        if batch_size is None:
            batch_size = (int(voxel_coors[:, 0].max().item()) + 1) if voxel_coors.numel() else 1
        return int(batch_size) * per_batch

    def forward(self, voxel_feats, voxel_coors, batch_size=None):
        self.set_drop_info()
        voxel_coors = voxel_coors.long()
        self._key_bound = self._window_id_bound(voxel_coors, batch_size)
        if self.shuffle_voxels:
            shuffle_inds = torch.randperm(len(voxel_feats), device=voxel_feats.device)
            voxel_feats, voxel_coors = voxel_feats[shuffle_inds], voxel_coors[shuffle_inds]
        info = self.window_partition(voxel_coors)
        info['voxel_feats'], info['voxel_coors'] = voxel_feats, voxel_coors
        info = self.drop_voxel(info, 2)
        voxel_feats, voxel_coors = info['voxel_feats'], info['voxel_coors']
        for i in range(2):
            info[f'flat2win_inds_shift{i}'] = get_flat2win_inds_v2(
                info[f'batch_win_inds_shift{i}'], info[f'voxel_drop_level_shift{i}'], self.drop_info, debug=self.debug,
                key_bound=self._key_bound)
            info[f'pos_dict_shift{i}'] = self.get_pos_embed(
                info[f'flat2win_inds_shift{i}'], info[f'coors_in_win_shift{i}'], voxel_feats.size(1), voxel_feats.dtype)
            info[f'key_mask_shift{i}'] = self.get_key_padding_mask(info[f'flat2win_inds_shift{i}'])
        if self.shuffle_voxels:
            info['shuffle_inds'] = shuffle_inds
        return info

    def drop_single_shift(self, batch_win_inds):
        """keep mask + drop level of every voxel from the population of its window (:128-148)."""
        # one group-rank pass gives the rank inside the window AND the window populations (the reference's bincount is a
        # second pass with a host read-back of its own); level and keep decision in one launch
        conti, inner, counts = group_rank(batch_win_inds, getattr(self, '_key_bound', None))   # (set by forward(); direct calls: no bound)
        n = batch_win_inds.numel()
        if not batch_win_inds.is_cuda:   # (CPU stand-ins of oracle/cpu_port.py)
            num_per_voxel = counts.to(batch_win_inds.dtype)[conti.long()]
            drop_lvl = -torch.ones_like(batch_win_inds)
            target = torch.zeros_like(batch_win_inds)
            for dl in self.drop_info:
                lower, upper = self.drop_info[dl]['drop_range']
                m = (num_per_voxel >= lower) & (num_per_voxel < upper)
                target[m] = self.drop_info[dl]['max_tokens']
                drop_lvl[m] = dl
            return inner.to(batch_win_inds.dtype) < target, drop_lvl
        import ctypes
        levels = list(self.drop_info)
        i64 = lambda v: (ctypes.c_int64 * len(v))(*[int(a) for a in v])
        keep = torch.empty((n,), dtype=torch.uint8, device=batch_win_inds.device)
        drop_lvl = torch.empty((n,), dtype=torch.int64, device=batch_win_inds.device)
        L.check(L.lib.ococc_sst_drop_level_i64(
            L.ptr(conti), L.ptr(inner), L.ptr(counts), n, len(levels), i64([self.drop_info[d]['drop_range'][0] for d in levels]),
            i64([min(self.drop_info[d]['drop_range'][1], 2 ** 62) for d in levels]),
            i64([self.drop_info[d]['max_tokens'] for d in levels]), i64(levels), L.ptr(keep), L.ptr(drop_lvl), L.stream()),
            'sst_drop_level')
        return keep.bool(), drop_lvl.to(batch_win_inds.dtype)

    def drop_voxel(self, info, num_shifts):
        """Two sequential drops: shift 0, then shift 1 on the survivors (:150-220)."""
        win0 = info['batch_win_inds_shift0']
        n_all = win0.shape[0]
        keep_inds = torch.arange(n_all, device=win0.device, dtype=torch.long)
        keep0, lvl0 = self.drop_single_shift(win0)
        keep0 = torch.where(keep0)[0]          # (index lists: ONE read-back per mask instead of one per masked tensor)
        # (the list's length is on the host with it: a shift that drops nothing -- every window below its level's
        # max_tokens, the usual case for object grids -- selects every row, and the selections are skipped: for 260 k
        # voxels the feature gather, its index_add backward and a dozen index gathers, ~0.3 ms per step)
        win1 = info['batch_win_inds_shift1']
        if keep0.numel() != n_all:
            lvl0, keep_inds, win0, win1 = lvl0[keep0], keep_inds[keep0], win0[keep0], win1[keep0]
        keep1, lvl1 = self.drop_single_shift(win1)
        keep1 = torch.where(keep1)[0]
        if keep1.numel() != win1.shape[0]:
            keep_inds, lvl0, win0, lvl1, win1 = keep_inds[keep1], lvl0[keep1], win0[keep1], lvl1[keep1], win1[keep1]
        info['voxel_keep_inds'] = keep_inds
        info['voxel_drop_level_shift0'], info['batch_win_inds_shift0'] = lvl0, win0
        info['voxel_drop_level_shift1'], info['batch_win_inds_shift1'] = lvl1, win1
        keep = info['voxel_keep_inds']
        if keep.numel() == n_all:
            return info
        for k, v in list(info.items()):
            if isinstance(v, torch.Tensor) and len(v) == n_all and k not in (
                    'voxel_keep_inds', 'voxel_drop_level_shift0', 'batch_win_inds_shift0',
                    'voxel_drop_level_shift1', 'batch_win_inds_shift1'):
                # (index_select: for the features its backward is one index_add launch -- that of v[keep] sorts the indices
                # first, ~50 launches and 0.45 ms per step for 260 k voxels)
                info[k] = v.index_select(0, keep)
        return info

    @torch.no_grad()
    def window_partition(self, coors):
        info = {}
        if coors.is_cuda and coors.dtype == torch.int64:
            both = get_window_coors_both(coors, self.sparse_shape, self.window_shape)   # one launch for the two shifts
        else:
            both = [get_window_coors(coors, self.sparse_shape, self.window_shape, i == 1) for i in range(2)]
        for i in range(2):
            info[f'batch_win_inds_shift{i}'], info[f'coors_in_win_shift{i}'] = both[i]
        return info

    @torch.no_grad()
    def get_pos_embed(self, inds_dict, coors_in_win, feat_dim, dtype):
        """Sinusoidal embedding of the in-window coordinate (:239-305)."""
        ws = self.window_shape
        if len(ws) == 2 or ws[-1] == 1:
            ndim, (win_x, win_y), win_z = 2, ws[:2], 0
        else:
            ndim = 3
            win_x, win_y, win_z = ws
        z, y, x = coors_in_win[:, 0] - win_z / 2, coors_in_win[:, 1] - win_y / 2, coors_in_win[:, 2] - win_x / 2
        if self.normalize_pos:
            x, y, z = x / win_x * 2 * 3.1415, y / win_y * 2 * 3.1415, z / win_z * 2 * 3.1415
        pos_length = feat_dim // ndim
        key = (pos_length, float(self.pos_temperature), str(coors_in_win.device))
        if getattr(self, '_inv_freq_key', None) != key:   # (the frequency table: the reference's own expression, built once)
            inv = torch.arange(pos_length, dtype=torch.float32, device=coors_in_win.device)
            self._inv_freq, self._inv_freq_key = self.pos_temperature ** (2 * (inv // 2) / pos_length), key
        inv_freq = self._inv_freq
        if (coors_in_win.is_cuda and coors_in_win.dtype == torch.int64 and dtype in (torch.float32, torch.bfloat16)
                and pos_length % 2 == 0):
            import ctypes
            ciw = coors_in_win.contiguous()
            win3 = (ctypes.c_int32 * 3)(int(win_x), int(win_y), int(win_z))
            made = {}

            def pos_of(dt):
                # the table in the dtype its reader wants (the kernel rounds once): the bf16 encoder layers ask for bf16 and
                # the f32 table -- 33 M sines and cosines at configs[4]'s size -- is only built if somebody reads it
                if dt not in made:
                    pos = torch.empty((ciw.size(0), feat_dim), dtype=dt, device=ciw.device)
                    L.check(L.lib.ococc_sst_pos_embed(L.ptr(ciw), ciw.size(0), win3, ndim, int(bool(self.normalize_pos)),
                                                      L.ptr(inv_freq), pos_length, feat_dim, L.ptr(pos), L.dtype_code(dt),
                                                      L.stream()), 'sst_pos_embed')
                    made[dt] = pos
                return made[dt]
            inds_dict['_ococc_pos_fn'] = pos_of   # flat token order: what the fused encoder layers read
            return LazyWindowDict(lambda: flat2window_v2(pos_of(dtype), inds_dict))
        emb = []
        for a in ([x, y, z] if ndim == 3 else [x, y]):
            e = a[:, None] / inv_freq[None, :]
            emb.append(torch.stack([e[:, ::2].sin(), e[:, 1::2].cos()], dim=-1).flatten(1))
        pos = torch.cat(emb, dim=-1).to(dtype)
        gap = feat_dim - pos.size(1)
        if gap > 0:
            pos = torch.cat([pos, torch.zeros((pos.size(0), gap), dtype=dtype, device=pos.device)], dim=1)
        inds_dict['_ococc_pos_fn'] = lambda dt: pos.to(dt)   # flat token order: what the fused encoder layers read
        return LazyWindowDict(lambda: flat2window_v2(pos, inds_dict))

    @torch.no_grad()
    def get_key_padding_mask(self, ind_dict):
        def build():
            n = len(ind_dict['voxel_drop_level'])
            ones = torch.ones((n, 1), device=ind_dict['voxel_drop_level'].device).bool()
            d = flat2window_v2(ones, ind_dict)
            return {k: v.logical_not().squeeze(2) for k, v in d.items()}
        return LazyWindowDict(build)


_attn_probe = None  # measurement hook (bench.py --workload sst): .wrap(T, launch) times the forward kernel


def set_attn_probe(probe):
    global _attn_probe
    _attn_probe = probe


class _WindowAttnCore(torch.autograd.Function):
    """softmax(q k^T / sqrt(d) + mask) v on padded windows, bf16 MFMA kernel."""

    @staticmethod
    def forward(ctx, q, k, v, key_len, num_heads):
        nW, T, C = q.shape
        D = C // num_heads
        qb, kb, vb = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
        out = torch.empty_like(qb)
        lse = torch.empty((nW, num_heads, T), dtype=torch.float32, device=q.device)
        scale = float(D) ** -0.5
        def launch():
            L.check(L.lib.ococc_window_attn_fwd_bf16(L.ptr(qb), L.ptr(kb), L.ptr(vb), C, C, C, L.ptr(key_len), nW, T,
                                                     num_heads, D, scale, L.ptr(out), C, L.ptr(lse), L.stream()),
                    'window_attn_fwd')
        if _attn_probe is not None:
            _attn_probe.wrap(nW, T, num_heads, D, launch)
        else:
            launch()
        ctx.save_for_backward(qb, kb, vb, out, lse, key_len)
        ctx.meta = (num_heads, D, scale, q.dtype)
        return out.to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        qb, kb, vb, out, lse, key_len = ctx.saved_tensors
        H, D, scale, dt = ctx.meta
        nW, T, C = qb.shape
        do = dout.to(torch.bfloat16).contiguous()
        dq, dk, dv = torch.empty_like(qb), torch.empty_like(kb), torch.empty_like(vb)
        L.check(L.lib.ococc_window_attn_bwd_bf16(L.ptr(qb), L.ptr(kb), L.ptr(vb), C, C, C, L.ptr(out), L.ptr(do), C,
                                                 L.ptr(lse), L.ptr(key_len), nW, T, H, D, scale, L.ptr(dq), L.ptr(dk),
                                                 L.ptr(dv), C, C, C, L.stream()), 'window_attn_bwd')
        return dq.to(dt), dk.to(dt), dv.to(dt), None, None


class _WindowAttnPacked(torch.autograd.Function):
    """Same kernel on a packed bf16 [nW, T, 3E] tensor (q | k | v per token, row stride 3E): no
    per-operand copies, and the backward writes dq | dk | dv into one packed tensor."""

    @staticmethod
    def forward(ctx, qkv, key_len, num_heads):
        nW, T, C3 = qkv.shape
        E = C3 // 3
        D = E // num_heads
        assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
        out = torch.empty((nW, T, E), dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty((nW, num_heads, T), dtype=torch.float32, device=qkv.device)
        scale = float(D) ** -0.5
        base = qkv.data_ptr()

        def launch():
            L.check(L.lib.ococc_window_attn_fwd_bf16(base, base + 2 * E, base + 4 * E, C3, C3, C3, L.ptr(key_len), nW,
                                                     T, num_heads, D, scale, L.ptr(out), E, L.ptr(lse), L.stream()),
                    'window_attn_fwd')
        if _attn_probe is not None:
            _attn_probe.wrap(nW, T, num_heads, D, launch)
        else:
            launch()
        ctx.save_for_backward(qkv, out, lse, key_len)
        ctx.meta = (num_heads, D, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, key_len = ctx.saved_tensors
        H, D, scale = ctx.meta
        nW, T, C3 = qkv.shape
        E = C3 // 3
        do = dout.to(torch.bfloat16).contiguous()
        dqkv = torch.empty_like(qkv)
        b, g = qkv.data_ptr(), dqkv.data_ptr()
        L.check(L.lib.ococc_window_attn_bwd_bf16(b, b + 2 * E, b + 4 * E, C3, C3, C3, L.ptr(out), L.ptr(do), E,
                                                 L.ptr(lse), L.ptr(key_len), nW, T, H, D, scale, g, g + 2 * E,
                                                 g + 4 * E, C3, C3, C3, L.stream()), 'window_attn_bwd')
        return dqkv, None, None


class _WindowAttnFlat(torch.autograd.Function):
    """Attention over all drop levels on the flat packed [V, 3E] token tensor: the kernels gather each
    window's tokens through a slot -> row index (ococc_window_attn_*_gather_bf16), so no padded window copy is
    built and the output / gradients are written straight in token order."""

    @staticmethod
    def forward(ctx, qkv, num_heads, *level_args):
        levels = [level_args[i:i + 4] for i in range(0, len(level_args), 4)]   # (tok, key_len, nW, T)
        out, lses, meta = _attn_flat_forward(qkv, num_heads, levels)
        ctx.save_for_backward(qkv, out, *lses, *[t for lv in levels for t in lv[:2]])
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, dout):
        n = len(ctx.meta[3])
        saved = ctx.saved_tensors
        qkv, out, lses, rest = saved[0], saved[1], saved[2:2 + n], saved[2 + n:]
        dqkv = _attn_flat_backward(qkv, out, dout, lses, rest, ctx.meta)
        return (dqkv, None) + (None,) * (4 * n)


def _attn_flat_forward(qkv, num_heads, levels):
    """the gather kernels over every drop level: (out [V, E] bf16, log-sum-exp per level, meta for the backward)"""
    V, C3 = qkv.shape
    E = C3 // 3
    D = E // num_heads
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    out = torch.empty((V, E), dtype=torch.bfloat16, device=qkv.device)
    scale = float(D) ** -0.5
    b = qkv.data_ptr()
    lses = []
    for tok, key_len, nW, T in levels:
        lse = torch.empty((nW, num_heads, T), dtype=torch.float32, device=qkv.device)

        def launch():
            L.check(L.lib.ococc_window_attn_fwd_gather_bf16(b, b + 2 * E, b + 4 * E, C3, C3, C3, L.ptr(tok),
                                                            L.ptr(key_len), nW, T, num_heads, D, scale,
                                                            L.ptr(out), E, L.ptr(lse), L.stream()),
                    'window_attn_fwd_gather')
        if _attn_probe is not None:
            _attn_probe.wrap(nW, T, num_heads, D, launch)
        else:
            launch()
        lses.append(lse)
    return out, lses, (num_heads, D, scale, [(lv[2], lv[3]) for lv in levels])


def _attn_flat_backward(qkv, out, dout, lses, tok_and_len, meta):
    """d(q | k | v) [V, 3E] bf16 of _attn_flat_forward; ``tok_and_len`` = (tok, key_len) of every level, flattened"""
    H, D, scale, shapes = meta
    V, C3 = qkv.shape
    E = C3 // 3
    do = dout.to(torch.bfloat16).contiguous()
    dqkv = torch.empty_like(qkv)   # every token belongs to exactly one window of one level: fully written
    b, g = qkv.data_ptr(), dqkv.data_ptr()
    for i, (nW, T) in enumerate(shapes):
        tok, key_len = tok_and_len[2 * i], tok_and_len[2 * i + 1]
        L.check(L.lib.ococc_window_attn_bwd_gather_bf16(b, b + 2 * E, b + 4 * E, C3, C3, C3, L.ptr(out), L.ptr(do), E,
                                                        L.ptr(lses[i]), L.ptr(tok), L.ptr(key_len), nW, T, H, D,
                                                        scale, g, g + 2 * E, g + 4 * E, C3, C3, C3, L.stream()),
                'window_attn_bwd_gather')
    return dqkv


class _BigWindowBlock(torch.autograd.Function):
    """The attention block (in-projection, per-window attention kernels, out-projection, residual sum, LayerNorm:
    sst_basic_block_v2.py:41-75, 105-118) of the rows whose windows have more tokens than the 64-slot tiles of
    csrc/window_block.hip hold.  Forward: the bf16 operator chain (library GEMMs with bf16 biases, csrc/window_attn.hip,
    a bf16 residual sum, the stand-alone LayerNorm kernel).  Backward: the chain and store points of the fused block
    kernels' backward (window_attn_block_bwd_kernel) -- d z1 in f32, rounded once as the operand of the two products
    that consume it, f32 accumulation of every parameter gradient -- written out on the few rows concerned, so that these
    rows' gradients carry the same roundings as everybody else's (and one oracle, oracle/sst_ref.py
    encoder_layer_backward, describes both).  Autograd over the bf16 operators rounded the gradient at every operator
    boundary and every weight-gradient slab to bf16: 3e-3 norm-wise on these rows against 7e-4 on the others."""

    @staticmethod
    def forward(ctx, xb, posb, w_in, b_in, w_out, b_out, gamma, beta, eps, num_heads, *level_args):
        dt = torch.bfloat16
        E = xb.shape[1]
        levels = [level_args[i:i + 4] for i in range(0, len(level_args), 4)]
        w16, b16 = w_in.to(dt), b_in.to(dt)
        xp = xb + posb
        qkv = torch.empty((xb.shape[0], 3 * E), dtype=dt, device=xb.device)
        torch.addmm(b16[:2 * E], xp, w16[:2 * E].t(), out=qkv[:, :2 * E])
        torch.addmm(b16[2 * E:], xb, w16[2 * E:].t(), out=qkv[:, 2 * E:])
        o, lses, meta = _attn_flat_forward(qkv, num_heads, levels)
        wo16 = w_out.to(dt)
        z1 = xb + torch.nn.functional.linear(o, wo16, b_out.to(dt))
        y1 = layer_norm_act(z1, gamma, beta, eps, 'none')
        ctx.save_for_backward(xb, xp, qkv, o, z1, w16, wo16, gamma, *lses, *[t for lv in levels for t in lv[:2]])
        ctx.meta, ctx.eps = meta, eps
        return y1

    @staticmethod
    def backward(ctx, dy1):
        n = len(ctx.meta[3])
        t = ctx.saved_tensors
        xb, xp, qkv, o, z1, w16, wo16, gamma = t[:8]
        lses, rest = t[8:8 + n], t[8 + n:]
        E = xb.shape[1]
        r16 = lambda v: v.to(torch.bfloat16)
        # LayerNorm backward on the f32 statistics of the stored (bf16) sums
        z = z1.float()
        mean = z.mean(1, keepdim=True)
        zc = z - mean
        rstd = torch.rsqrt((zc * zc).mean(1, keepdim=True) + ctx.eps)
        xh = zc * rstd
        dy = dy1.float()
        dg = dy * gamma.float()
        dz1 = (dg - dg.mean(1, keepdim=True) - xh * (dg * xh).mean(1, keepdim=True)) * rstd
        g_gamma, g_beta = (dy * xh).sum(0), dy.sum(0)
        dz1r = r16(dz1)                                             # operand of the out-projection's two products
        do = _mm_f32(dz1r, wo16)
        g_wo = _mm_f32(dz1r.t(), o)
        g_bo = dz1r.float().sum(0)
        dqkv = _attn_flat_backward(qkv, o, r16(do), lses, rest, ctx.meta)
        dx = r16(_mm_f32(dqkv, w16) + dz1)
        g_w = torch.cat([_mm_f32(dqkv[:, :2 * E].t(), xp), _mm_f32(dqkv[:, 2 * E:].t(), xb)], 0)
        g_b = dqkv.float().sum(0)
        return (dx, None, g_w, g_b, g_wo, g_bo, g_gamma, g_beta, None, None) + (None,) * (4 * n)


try:   # bf16 operands, f32 result without a rounding in between (torch >= 2.8); else f32 copies of the operands
    torch.mm(torch.zeros(1, 1, dtype=torch.bfloat16), torch.zeros(1, 1, dtype=torch.bfloat16), out_dtype=torch.float32)
    _MM_OUT_DTYPE = True
except (TypeError, RuntimeError):
    _MM_OUT_DTYPE = False


def _mm_f32(a, b):
    """a @ b for bf16 operands with f32 accumulation AND an f32 result"""
    if _MM_OUT_DTYPE and a.is_cuda:
        return torch.mm(a.contiguous(), b.contiguous(), out_dtype=torch.float32)
    return a.float() @ b.float()


class _TokenLinear(torch.autograd.Function):
    """y = x W^T + b over ~1e5..1e6 token rows with a tiny [out, in] weight.  The weight gradient
    dW = dY^T X contracts over the token dimension; the BLAS heuristics give that one output tile
    per 64x128 of dW (a dozen workgroups on a 256-CU part, ~0.6 ms).  Here the token dimension is cut
    into 64 slabs, a batched GEMM produces 64 partial dW (thousands of tiles), and the partials are
    summed in f32."""
    SLABS = 64

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w if ctx.needs_input_grad[0] else None
        n, s = x.shape[0], _TokenLinear.SLABS
        m = (n // s) * s
        dw = None
        if m:
            part = torch.bmm(dy[:m].view(s, m // s, -1).transpose(1, 2), x[:m].view(s, m // s, -1))
            dw = part.float().sum(0)
        if m < n:
            tail = (dy[m:].t() @ x[m:]).float()
            dw = tail if dw is None else dw + tail
        db = dy.sum(0, dtype=torch.float32).to(dy.dtype) if ctx.has_bias else None  # f32 accumulation, no f32 copy
        return dx, dw.to(w.dtype), db


class _QkvProjection(torch.autograd.Function):
    """The in-projection of a window-attention layer on flat tokens: q | k from x + pos, v from x
    (sst_basic_block_v2.py:41-75: ``q = k = src + pos; v = src`` into nn.MultiheadAttention), written by two GEMMs
    straight into the column slices of ONE [V, 3E] buffer -- the layout the attention kernels read -- instead of two
    results concatenated (a 150 us copy per layer at 260 k tokens, and the matching split copies in the backward).
    Weight gradients as in _TokenLinear (token dimension cut into slabs), read from the column slices in place."""

    @staticmethod
    def forward(ctx, x, pos, w, b):
        E = x.shape[1]
        xp = x + pos
        out = torch.empty((x.shape[0], 3 * E), dtype=x.dtype, device=x.device)
        torch.addmm(b[:2 * E], xp, w[:2 * E].t(), out=out[:, :2 * E])
        torch.addmm(b[2 * E:], x, w[2 * E:].t(), out=out[:, 2 * E:])
        ctx.save_for_backward(x, xp, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, xp, w = ctx.saved_tensors
        E = x.shape[1]
        g = g.contiguous()
        g_qk, g_v = g[:, :2 * E], g[:, 2 * E:]
        dx = dpos = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            d_xp = g_qk @ w[:2 * E]
            dpos = d_xp if ctx.needs_input_grad[1] else None
            dx = torch.addmm(d_xp, g_v, w[2 * E:]) if ctx.needs_input_grad[0] else None
        n, s = x.shape[0], _TokenLinear.SLABS
        m = (n // s) * s
        dw = torch.zeros((3 * E, E), dtype=torch.float32, device=x.device)
        if m:
            gs = g[:m].view(s, m // s, 3 * E)
            dw[:2 * E] += torch.bmm(gs[:, :, :2 * E].transpose(1, 2), xp[:m].view(s, m // s, E)).float().sum(0)
            dw[2 * E:] += torch.bmm(gs[:, :, 2 * E:].transpose(1, 2), x[:m].view(s, m // s, E)).float().sum(0)
        if m < n:
            dw[:2 * E] += (g_qk[m:].t() @ xp[m:]).float()
            dw[2 * E:] += (g_v[m:].t() @ x[m:]).float()
        db = g.sum(0, dtype=torch.float32).to(g.dtype)
        return dx, dpos, dw.to(w.dtype), db


class _ScatterRows(torch.autograd.Function):
    """out[slot[i]] = feat[pos[i]] into a zero [rows, C] tensor; slot and pos are injective, so the
    backward is the mirror gather (no sort-based index_put backward)."""

    @staticmethod
    def forward(ctx, feat, pos, slot, rows):
        out = feat.new_zeros((rows, feat.shape[1]))
        out.index_copy_(0, slot, feat.index_select(0, pos))
        ctx.save_for_backward(pos, slot)
        ctx.n = feat.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad):
        pos, slot = ctx.saved_tensors
        g = grad.new_zeros((ctx.n, grad.shape[1]))
        g.index_copy_(0, pos, grad.index_select(0, slot))
        return g, None, None, None


def _window_maps(ind_dict, key_padding_dict):
    """Per drop level (slot, pos, num_windows, max_tokens, key_len), cached on the index dict."""
    maps = ind_dict.get('_ococc_maps')
    if maps is None:
        maps = {}
        info = ind_dict['batching_info']
        pop = ind_dict.get('_ococc_populations')
        for dl in info:
            if dl not in ind_dict:
                continue
            slot, flat_pos = ind_dict[dl]
            if pop is not None and dl in pop:   # window populations straight from the group-rank kernel
                key_len, T = pop[dl], info[dl]['max_tokens']
                nW = int(key_len.numel())
            else:
                mask = key_padding_dict[dl]
                nW, T = mask.shape
                key_len = (~mask).sum(1).to(torch.int32)
            tok = torch.full((nW * T,), -1, dtype=torch.int32, device=slot.device)
            tok[slot] = flat_pos[0].to(torch.int32)
            maps[dl] = (slot, flat_pos[0], nW, T, key_len, tok)
        ind_dict['_ococc_maps'] = maps
    return maps


BIG_WINDOW_BLOCK = True      # False: the rows of windows above 64 tokens through autograd over the bf16 operators (_BigWindowBlock)
FUSED_ENCODER_LAYER = True   # False: the per-operator flat path (GEMMs through torch + the per-window attention kernels)


def _fused_maps(ind_dict, pos_dict, key_padding_dict, num_tokens, dtype):
    """(TilePlan of the drop levels whose windows fit a 64-slot tile, (rows, maps) of the larger ones or None,
    positional embedding in flat token order), cached on the index dict of the shift."""
    hit = ind_dict.get('_ococc_fused')
    if hit is None:
        from .fused_block import TILE, TilePlan
        maps = _window_maps(ind_dict, key_padding_dict)
        pos_flat = ind_dict.get('_ococc_pos_flat')
        if pos_flat is None:
            fn = ind_dict.get('_ococc_pos_fn')
            pos_flat = (fn(dtype) if fn is not None else window2flat_v2(pos_dict, ind_dict).to(dtype)).contiguous()
            ind_dict['_ococc_pos_flat'] = pos_flat
        small = [(tok, key_len, nW, T) for (slot, pos, nW, T, key_len, tok) in maps.values() if T <= TILE]
        large = {dl: m for dl, m in maps.items() if m[3] > TILE and m[2] > 0 and m[1].numel() > 0}
        device = pos_flat.device
        plan = TilePlan(small, device)
        big = None
        if large:
            rows = torch.cat([m[1] for m in large.values()])            # flat rows of the large windows' tokens
            inv = torch.full((num_tokens,), -1, dtype=torch.int32, device=device)
            inv[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=device)
            remapped = {}
            for dl, (slot, pos, nW, T, key_len, tok) in large.items():
                t2 = torch.where(tok >= 0, inv[tok.clamp(min=0).long()], tok)
                remapped[dl] = (slot, inv[pos].long(), nW, T, key_len, t2)
            big = (rows, remapped)
        hit = ind_dict['_ococc_fused'] = (plan, big, pos_flat)
    return hit


class WindowMultiheadAttention(nn.Module):
    """Parameter layout of nn.MultiheadAttention (in_proj_weight [3E,E], in_proj_bias,
    out_proj.{weight,bias}); batch-first padded windows [nW, T, E]; attention core on the HIP kernel."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, cosine=False, tau_min=0.01, non_shared_tau=False):
        super().__init__()
        if dropout != 0:
            raise NotImplementedError('attention-probability dropout is not built into the window-attention kernel; the '
                                      'reference SST configs use dropout=0.0 (sst_basic_block_v2.py:41-75 passes it to '
                                      'nn.MultiheadAttention)')
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)
        # scaled cosine attention (CosineMultiheadAttention, cosine_msa.py:123-185,449-466): per-head unit
        # vectors, logits = cos / clamp(tau, tau_min); tau shared ([1,1,1]) or one per head ([1,H,1,1])
        self.tau_min = tau_min
        self.tau = (nn.Parameter(torch.ones(1, num_heads, 1, 1) if non_shared_tau else torch.ones(1, 1, 1))
                    if cosine else None)

    def _cosine_q_k(self, q, k):
        """q, k [..., E] -> unit vectors per head, q additionally divided by tau and multiplied by sqrt(d) so
        that the kernel's fixed 1/sqrt(d) scale leaves cos / tau."""
        H = self.num_heads
        D = self.embed_dim // H
        shp = q.shape
        qh = torch.nn.functional.normalize(q.reshape(-1, H, D).float(), dim=-1)
        kh = torch.nn.functional.normalize(k.reshape(-1, H, D).float(), dim=-1)
        tau = self.tau.clamp(min=self.tau_min).reshape(1, -1, 1).float()   # [1,1,1] or [1,H,1]
        qh = qh * (float(D) ** 0.5 / tau)
        return qh.reshape(shp).to(q.dtype), kh.reshape(shp).to(k.dtype)

    def forward(self, qk_in, v_in, key_padding_mask):
        nW, T, E = qk_in.shape
        w, b = self.in_proj_weight, self.in_proj_bias
        qk = torch.addmm(b[:2 * E], qk_in.reshape(nW * T, E), w[:2 * E].t())
        v = torch.addmm(b[2 * E:], v_in.reshape(nW * T, E), w[2 * E:].t())
        key_len = (~key_padding_mask).sum(1).to(torch.int32)  # valid tokens are a prefix of each window
        q_, k_ = qk[:, :E], qk[:, E:]
        if self.tau is not None:
            q_, k_ = self._cosine_q_k(q_, k_)
        o = _WindowAttnCore.apply(q_.reshape(nW, T, E), k_.reshape(nW, T, E), v.view(nW, T, E),
                                  key_len, self.num_heads)
        return self.out_proj(o.reshape(nW * T, E)).view(nW, T, E)

    def forward_flat(self, x, pos_flat, maps, dtype):
        """MI355X form of the same layer: the q/k/v and output projections are per-token, so they
        run on the V real tokens (not on the ~3x larger padded windows) as bf16 GEMMs; one packed
        [V, 3E] tensor is scattered into the padded window layout per drop level, the attention core
        reads it in place, and its output is gathered straight back to token order."""
        E, H = self.embed_dim, self.num_heads
        w, b = self.in_proj_weight.to(dtype), self.in_proj_bias.to(dtype)
        x16 = x.to(dtype)
        if self.tau is not None:
            qk = _TokenLinear.apply(x16 + pos_flat, w[:2 * E], b[:2 * E])
            v = _TokenLinear.apply(x16, w[2 * E:], b[2 * E:])
            q_, k_ = self._cosine_q_k(qk[:, :E], qk[:, E:])
            qkv = torch.cat([q_, k_, v], 1)
        else:
            qkv = _QkvProjection.apply(x16, pos_flat.to(dtype), w, b)   # q | k | v in one buffer, no concatenation
        covered = sum(int(m[0].numel()) for m in maps.values())
        if covered == x.shape[0]:   # every token sits in a window (always, after drop_voxel): gather kernels
            args = []
            for dl, (slot, pos, nW, T, key_len, tok) in maps.items():
                args += [tok, key_len, nW, T]
            o_flat = _WindowAttnFlat.apply(qkv, H, *args)
        else:
            o_flat = None
            for dl, (slot, pos, nW, T, key_len, tok) in maps.items():
                packed = _ScatterRows.apply(qkv, pos, slot, nW * T).view(nW, T, 3 * E)
                o = _WindowAttnPacked.apply(packed, key_len, H).view(nW * T, E)
                part = _ScatterRows.apply(o, slot, pos, x.shape[0])
                o_flat = part if o_flat is None else o_flat + part
        return _TokenLinear.apply(o_flat, self.out_proj.weight.to(dtype), self.out_proj.bias.to(dtype))


class WindowAttention(nn.Module):

    def __init__(self, d_model, nhead, dropout, batch_first=False, layer_id=None, layer_cfg=dict()):
        super().__init__()
        assert not layer_cfg.get('linear', False), 'the linear attention variant raises NotImplementedError upstream too'
        self.nhead = nhead
        self.self_attn = WindowMultiheadAttention(d_model, nhead, dropout=dropout, cosine=layer_cfg.get('cosine', False),
                                                  tau_min=layer_cfg.get('tau_min', 0.01),
                                                  non_shared_tau=layer_cfg.get('non_shared_tau', False))
        self.layer_id = layer_id
        self.compute_dtype = layer_cfg.get('compute_dtype', None)  # torch.bfloat16 -> flat-token bf16 path

    def forward(self, feat_2d, pos_dict, ind_dict, key_padding_dict):
        if self.compute_dtype is not None:
            maps = _window_maps(ind_dict, key_padding_dict)
            pos_flat = ind_dict.get('_ococc_pos_flat')
            if pos_flat is None:
                pos_flat = window2flat_v2(pos_dict, ind_dict).to(self.compute_dtype)
                ind_dict['_ococc_pos_flat'] = pos_flat
            return self.self_attn.forward_flat(feat_2d, pos_flat, maps, self.compute_dtype)
        feat_3d_dict = flat2window_v2(feat_2d, ind_dict)
        out = {}
        for name, feat_3d in feat_3d_dict.items():
            pos = pos_dict[name]
            qk = feat_3d + pos if pos is not None else feat_3d
            out[name] = self.self_attn(qk, feat_3d, key_padding_dict[name])
        return window2flat_v2(out, ind_dict)


def _activation(name):
    return {'relu': torch.nn.functional.relu, 'gelu': torch.nn.functional.gelu}[name]


class EncoderLayer(nn.Module):

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', batch_first=False,
                 layer_id=None, mlp_dropout=0, layer_cfg=dict()):
        super().__init__()
        self.win_attn = WindowAttention(d_model, nhead, dropout, layer_id=layer_id, layer_cfg=layer_cfg)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(mlp_dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.use_bn = layer_cfg.get('use_bn', False)
        if self.use_bn:  # sst_basic_block_v2.py:90-93
            from ..registry import build_norm_layer
            mom = layer_cfg.get('mom', 0.1)
            self.norm1 = build_norm_layer(dict(type='naiveSyncBN1d', momentum=mom), d_model)[1]
            self.norm2 = build_norm_layer(dict(type='naiveSyncBN1d', momentum=mom), d_model)[1]
        else:
            self.norm1, self.norm2 = nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2 = nn.Dropout(mlp_dropout), nn.Dropout(mlp_dropout)
        self.activation = _activation(activation)
        self._act_name = activation
        self.post_norm = layer_cfg.get('post_norm', True)
        self.compute_dtype = layer_cfg.get('compute_dtype', None)

    @staticmethod
    def _ln(norm, x):
        if not isinstance(norm, nn.LayerNorm):  # use_bn: batch norm over the tokens
            return norm(x.float()).to(x.dtype)
        return layer_norm_act(x, norm.weight, norm.bias, norm.eps, 'none')

    def _ffn(self, x):
        if self.compute_dtype is None:
            return self.linear2(self.dropout(self.activation(self.linear1(x))))
        dt, lin = self.compute_dtype, _TokenLinear.apply
        h = self.activation(lin(x.to(dt), self.linear1.weight.to(dt), self.linear1.bias.to(dt)))
        return lin(self.dropout(h), self.linear2.weight.to(dt), self.linear2.bias.to(dt))

    def _fusable(self):
        """The tile kernels of csrc/window_block.hip cover the reference SST configuration (d_model 128, 8 heads, ffn
        256, LayerNorm, post-norm, no dropout, softmax attention): anything else keeps the per-operator path."""
        mha = self.win_attn.self_attn
        drop = self.training and (self.dropout.p > 0 or self.dropout1.p > 0 or self.dropout2.p > 0)
        wanted = FUSED_ENCODER_LAYER and self.compute_dtype == torch.bfloat16
        ok = (wanted and self.post_norm and not self.use_bn
              and mha.tau is None and mha.embed_dim == 128 and mha.num_heads == 8 and self.linear1.out_features == 256
              and self._act_name in ('gelu', 'relu') and not drop)
        if wanted and not ok:
            from .. import _lib as L
            L.log_once(('sst-layer', mha.embed_dim, mha.num_heads, self.linear1.out_features, self._act_name, self.post_norm,
                        self.use_bn, mha.tau is None, bool(drop)),
                       f'SST encoder layer (d_model {mha.embed_dim}, {mha.num_heads} heads, ffn {self.linear1.out_features}, '
                       f'{self._act_name}, post_norm={self.post_norm}, use_bn={self.use_bn}, cosine={mha.tau is not None}, '
                       f'dropout={bool(drop)}) is outside the fused block kernels (128 / 8 / 256, gelu|relu, post-norm '
                       'LayerNorm, softmax, no dropout): running operator by operator')
        return ok

    def _forward_fused(self, src, pos_dict, ind_dict, key_padding_mask_dict):
        from . import fused_block as fb
        mha = self.win_attn.self_attn
        dt = self.compute_dtype
        x = src.to(dt).contiguous()
        plan, big, pos_flat = _fused_maps(ind_dict, pos_dict, key_padding_mask_dict, x.shape[0], dt)
        covered = plan.tokens == x.shape[0]
        y1 = fb.AttnBlock.apply(x, pos_flat, plan, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                mha.out_proj.bias, self.norm1.weight, self.norm1.bias, self.norm1.eps, mha.num_heads,
                                covered)
        if big is not None:   # windows of more than 64 tokens: per-window kernels on the rows they own
            rows, maps = big
            xb = x.index_select(0, rows)
            covered_big = sum(int(m[0].numel()) for m in maps.values()) == xb.shape[0]
            if BIG_WINDOW_BLOCK and mha.tau is None and covered_big:
                args = []
                for dl, (slot, pos, nW, T, key_len, tok) in maps.items():
                    args += [tok, key_len, nW, T]
                y1b = _BigWindowBlock.apply(xb, pos_flat.index_select(0, rows), mha.in_proj_weight, mha.in_proj_bias,
                                            mha.out_proj.weight, mha.out_proj.bias, self.norm1.weight, self.norm1.bias,
                                            self.norm1.eps, mha.num_heads, *args)
            else:
                ob = mha.forward_flat(xb, pos_flat.index_select(0, rows), maps, dt)
                y1b = self._ln(self.norm1, xb + ob)
            y1 = y1.index_copy(0, rows, y1b)
        return fb.FfnBlock.apply(y1, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                 self.norm2.weight, self.norm2.bias, self.norm2.eps, self._act_name)

    def forward(self, src, pos_dict, ind_dict, key_padding_mask_dict):
        if self.compute_dtype is not None and self._fusable():
            return self._forward_fused(src, pos_dict, ind_dict, key_padding_mask_dict)
        if self.compute_dtype is not None:
            src = src.to(self.compute_dtype)  # bf16 residual stream; LN statistics and GEMM accumulation stay f32
            assert self.post_norm
            src = self._ln(self.norm1, src + self.dropout1(self.win_attn(src, pos_dict, ind_dict, key_padding_mask_dict)))
            return self._ln(self.norm2, src + self.dropout2(self._ffn(src)))
        if self.post_norm:
            src = self._ln(self.norm1, src + self.dropout1(self.win_attn(src, pos_dict, ind_dict, key_padding_mask_dict)))
            src2 = self.linear2(self.dropout(self.activation(self.linear1(src))))
            return self._ln(self.norm2, src + self.dropout2(src2))
        src = src + self.dropout1(self.win_attn(self._ln(self.norm1, src), pos_dict, ind_dict, key_padding_mask_dict))
        src2 = self.linear2(self.dropout(self.activation(self.linear1(self._ln(self.norm2, src)))))
        return src + self.dropout2(src2)


class BasicShiftBlockV2(nn.Module):
    """Two encoder layers, the second on the shifted windows."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', batch_first=False,
                 block_id=-100, layer_cfg=dict()):
        super().__init__()
        self.encoder_list = nn.ModuleList([
            EncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, batch_first,
                         layer_id=block_id * 2 + i, layer_cfg=layer_cfg) for i in range(2)])

    def forward(self, src, pos_dict_list, ind_dict_list, key_mask_dict_list, using_checkpoint=False):
        num_shifts = len(pos_dict_list)
        assert num_shifts in (1, 2)
        out = src
        for i in range(2):
            j = i % num_shifts
            out = self.encoder_list[i](out, pos_dict_list[j], ind_dict_list[j], key_mask_dict_list[j])
        return out


@BACKBONES.register_module()
class SSTv2(nn.Module):

    def __init__(self, d_model=[], nhead=[], num_blocks=6, dim_feedforward=[], dropout=0.0, activation='gelu',
                 output_shape=None, num_attached_conv=2, conv_in_channel=64, conv_out_channel=64,
                 norm_cfg=dict(type='naiveSyncBN2d', eps=1e-3, momentum=0.01), conv_cfg=dict(type='Conv2d', bias=False),
                 debug=True, in_channel=None, to_bev=True,
                 conv_kwargs=dict(kernel_size=3, dilation=2, padding=2, stride=1), checkpoint_blocks=[],
                 layer_cfg=dict(), conv_shortcut=False):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.checkpoint_blocks, self.conv_shortcut, self.to_bev = checkpoint_blocks, conv_shortcut, to_bev
        if in_channel is not None:
            self.linear0 = nn.Linear(in_channel, d_model[0])
        self.block_list = nn.ModuleList([
            BasicShiftBlockV2(d_model[i], nhead[i], dim_feedforward[i], dropout, activation, batch_first=False,
                              block_id=i, layer_cfg=layer_cfg) for i in range(num_blocks)])
        for name, p in self.named_parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.output_shape, self.debug, self.num_attached_conv = output_shape, debug, num_attached_conv
        if num_attached_conv > 0:
            convs = []
            for i in range(num_attached_conv):
                kw = conv_kwargs if isinstance(conv_kwargs, dict) else conv_kwargs[i]
                conv = build_conv_layer(conv_cfg, in_channels=conv_in_channel if i == 0 else conv_out_channel,
                                        out_channels=conv_out_channel, **kw)
                layers = [conv] + ([build_norm_layer(norm_cfg, conv_out_channel)[1]] if norm_cfg else []) + \
                    [nn.ReLU(inplace=True)]
                convs.append(nn.Sequential(*layers))
            self.conv_layer = nn.ModuleList(convs)

    def forward(self, voxel_info):
        num_shifts = 2
        assert voxel_info['voxel_coors'].dtype == torch.int64, 'data type of coors should be torch.int64!'
        ind_dicts = [voxel_info[f'flat2win_inds_shift{i}'] for i in range(num_shifts)]
        masks = [voxel_info[f'key_mask_shift{i}'] for i in range(num_shifts)]
        poss = [voxel_info[f'pos_dict_shift{i}'] for i in range(num_shifts)]
        out = voxel_info['voxel_feats']
        if hasattr(self, 'linear0'):
            out = self.linear0(out)
        for block in self.block_list:
            out = block(out, poss, ind_dicts, masks)
        if self.to_bev:
            batch_size = int(voxel_info['voxel_coors'][:, 0].max().item()) + 1   # (a read-back: only where the canvas needs it)
            out = self.recover_bev(out, voxel_info['voxel_coors'], batch_size)
        if self.num_attached_conv > 0:
            assert self.to_bev
            for conv in self.conv_layer:
                tmp = conv(out)
                out = tmp + out if (tmp.shape == out.shape and self.conv_shortcut) else tmp
        if not self.to_bev:
            out = {'voxel_feats': out, 'voxel_coors': voxel_info['voxel_coors']}
        return [out]

    def recover_bev(self, voxel_feat, coors, batch_size):
        """Scatter voxel rows into a dense [B, C, ny, nx] canvas (sst_v2.py:156-197)."""
        ny, nx = self.output_shape
        c = voxel_feat.shape[-1]
        canvas = torch.zeros((batch_size, ny * nx, c), dtype=voxel_feat.dtype, device=voxel_feat.device)
        canvas[coors[:, 0], coors[:, 2] * nx + coors[:, 3]] = voxel_feat
        return canvas.permute(0, 2, 1).reshape(batch_size, c, ny, nx)
