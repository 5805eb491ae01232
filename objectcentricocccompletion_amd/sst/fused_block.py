"""One SST encoder layer on the fused tile kernels of csrc/window_block.hip -- the MI355X form of
EncoderLayer.forward / WindowAttention.forward (mmdet3d/models/sst/sst_basic_block_v2.py:41-75,105-127, post-norm).

    y1 = norm1(x + out_proj(MHA(q = k = x + pos, v = x)))     ococc_window_attn_block_{fwd,bwd}_bf16
    y2 = norm2(y1 + linear2(act(linear1(y1))))                 ococc_token_ffn_block_{fwd,bwd}_bf16
    dW, db of the four linears                                 ococc_token_wgrad_bf16 + ococc_partial_rows_sum_f32

Windows of up to 64 tokens are packed into tiles of 64 slots (``TilePlan``, built once per batch and window shift by
ococc_window_tile_plan); windows of drop levels above 64 tokens keep the per-window attention kernels on the rows they
own (sst_modules.WindowMultiheadAttention.forward_flat) and both halves meet again in the FFN block, which does not
care about windows.  Nothing but the layer input is saved for backward: the backward kernels recompute."""
import ctypes
import os

import torch

from .. import _lib as L

TILE = 64
ACT = {'gelu': 0, 'relu': 1}
_probe = None   # measurement hook (bench.py --workload sst): .wrap(name, flops, launch) brackets a forward kernel with events


def set_probe(probe):
    global _probe
    _probe = probe


def _run(name, flops, launch):
    if _probe is not None:
        _probe.wrap(name, flops, launch)
    else:
        launch()


def _vp(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def _i64(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def linear_fragments(mats):
    """f32 matrices (2-D views, any strides: pass ``w.t()`` for the transposed operand) -> bf16 MFMA A-operand
    fragment tensors, all in one launch."""
    outs = []
    for i in range(0, len(mats), 16):
        part = mats[i:i + 16]
        for m in part:
            assert m.dim() == 2 and m.dtype == torch.float32 and m.shape[0] % 16 == 0 and m.shape[1] % 32 == 0
        dst = [torch.empty(m.shape[0] * m.shape[1], dtype=torch.bfloat16, device=m.device) for m in part]
        L.check(L.lib.ococc_linear_fragments_bf16(len(part), _vp([m.data_ptr() for m in part]),
                                                  _i64([m.shape[0] for m in part]), _i64([m.shape[1] for m in part]),
                                                  _i64([m.stride(0) for m in part]), _i64([m.stride(1) for m in part]),
                                                  _vp([d.data_ptr() for d in dst]), L.stream()), 'linear_fragments')
        outs += dst
    return outs


# training keeps the attention output + log-sum-exp of the attention block for its backward (OCOCC_SST_KEEP_ATTENTION=0: the
# backward kernel runs the attention forward again, rounds 3-4)
KEEP_ATTENTION = os.environ.get('OCOCC_SST_KEEP_ATTENTION', '1') == '1'


class TilePlan(object):
    """tile_rows / tile_span [num_tiles * 64] int32 (include/ococc_hip.h) for the windows of the given drop levels,
    ``tokens`` = how many token rows they cover."""

    def __init__(self, levels, device):
        # levels: [(tok [nW * T] int32 flat row per window slot or -1, key_len [nW] int32, nW, T)], T <= 64
        lens, offs, toks, base = [], [], [], 0
        for tok, key_len, nW, T in levels:
            assert T <= TILE
            lens.append(key_len.to(torch.int32))
            offs.append(base + torch.arange(nW, device=device, dtype=torch.int64) * T)
            toks.append(tok)
            base += nW * T
        n = sum(int(l.numel()) for l in lens)
        self.num_tiles, self.tokens, self.sum_sq = 0, 0, 0.0
        self.rows = self.span = None
        if n == 0:
            return
        win_len, win_off, tok = torch.cat(lens), torch.cat(offs), torch.cat(toks)
        self.rows = torch.empty(n * TILE, dtype=torch.int32, device=device)
        self.span = torch.empty(n * TILE, dtype=torch.int32, device=device)
        count = torch.zeros(1, dtype=torch.int32, device=device)
        nbytes = L.lib.ococc_window_tile_plan_workspace_bytes(n)
        ws = L.workspace(nbytes, device)
        L.check(L.lib.ococc_window_tile_plan(L.ptr(win_len), L.ptr(win_off), L.ptr(tok), n, TILE, n, L.ptr(self.rows),
                                             L.ptr(self.span), L.ptr(count), L.ptr(ws), nbytes, L.stream()),
                'window_tile_plan')
        # ONE read-back per batch and shift: tile count, token count, sum of squared populations (attention flops)
        stats = torch.stack([count[0].long(), win_len.sum(), (win_len.double() ** 2).sum().long()]).tolist()
        self.num_tiles, self.tokens, self.sum_sq = int(stats[0]), int(stats[1]), float(stats[2])
        self.rows, self.span = self.rows[:self.num_tiles * TILE], self.span[:self.num_tiles * TILE]


def _wgrad(items, num_tokens, device, ln_partial=None, ln_tiles=0):
    """items: [(G [V, ldg] bf16 view whose first n columns are used, n, X [V, k] bf16, xadd or None, add_rows)] ->
    [(dW [n, k] f32, db [n] f32)] through one wgrad launch and one row-sum launch.  ``ln_partial`` [ln_tiles, 2, 128] f32:
    the block's LayerNorm partial rows ride on the same row-sum launch; (d gamma, d beta) is appended to the result."""
    slabs = int(L.lib.ococc_token_wgrad_slabs(num_tokens))
    dwp = [torch.empty((slabs, n, x.shape[1]), dtype=torch.float32, device=device) for _, n, x, _, _ in items]
    dbp = [torch.empty((slabs, n), dtype=torch.float32, device=device) for _, n, _, _, _ in items]
    L.check(L.lib.ococc_token_wgrad_bf16(
        len(items), _vp([g.data_ptr() for g, *_ in items]), _i64([g.stride(0) for g, *_ in items]),
        _i64([n for _, n, *_ in items]), _vp([x.data_ptr() for _, _, x, _, _ in items]),
        _vp([None if a is None else a.data_ptr() for *_, a, _ in items]), _i64([r for *_, r in items]),
        _i64([x.shape[1] for _, _, x, _, _ in items]), num_tokens, slabs, _vp([t.data_ptr() for t in dwp]),
        _vp([t.data_ptr() for t in dbp]), L.stream()), 'token_wgrad')
    dw = [torch.empty(t.shape[1:], dtype=torch.float32, device=device) for t in dwp]
    db = [torch.empty(t.shape[1:], dtype=torch.float32, device=device) for t in dbp]
    src, dst, rows = dwp + dbp, dw + db, [slabs] * (len(dwp) + len(dbp))
    ln_out = None
    if ln_partial is not None and ln_tiles > 0:
        ln_out = torch.empty((2, 128), dtype=torch.float32, device=device)
        src, dst, rows = src + [ln_partial], dst + [ln_out], rows + [ln_tiles]
    L.check(L.lib.ococc_partial_rows_sum_f32(len(src), _vp([t.data_ptr() for t in src]), _i64(rows),
                                             _i64([t[0].numel() for t in src]), _vp([t.data_ptr() for t in dst]),
                                             L.stream()), 'partial_rows_sum')
    out = list(zip(dw, db))
    if ln_partial is not None:
        if ln_out is None:
            ln_out = torch.zeros((2, 128), dtype=torch.float32, device=device)
        out.append((ln_out[0], ln_out[1]))
    return out


class AttnBlock(torch.autograd.Function):
    """y1 rows of the plan's tokens; every other row of the result is zero (covered=False) or does not exist."""

    @staticmethod
    def forward(ctx, x, pos, plan, in_w, in_b, out_w, out_b, ln_w, ln_b, eps, num_heads, covered):
        V, E = x.shape
        assert x.dtype == torch.bfloat16 and x.is_contiguous() and (pos is None or (pos.dtype == x.dtype and pos.is_contiguous()))
        wqkv, wo = linear_fragments([in_w.detach(), out_w.detach()])
        bq, bo = in_b.detach().float().contiguous(), out_b.detach().float().contiguous()
        g1, b1 = ln_w.detach().float().contiguous(), ln_b.detach().float().contiguous()
        y = torch.empty_like(x) if covered else torch.zeros_like(x)
        # algorithmic flops on the real tokens: in-projection 2*3E*E, out-projection 2*E*E per token, attention 4*n*E per
        # token of an n-token window (what the reference's nn.MultiheadAttention computes without its padding)
        flops = plan.tokens * 8.0 * E * E + 4.0 * E * plan.sum_sq
        # a pass that will be differentiated keeps the attention output and the softmax's log-sum-exp (288 B per token): the
        # backward kernel reads them back instead of running the attention forward again (KEEP_ATTENTION = False: recompute)
        keep = KEEP_ATTENTION and any(ctx.needs_input_grad)
        o_save = lse_save = None
        if keep:
            o_save = (torch.empty if covered else torch.zeros)((V, E), dtype=torch.bfloat16, device=x.device)
            lse_save = torch.empty((V, num_heads), dtype=torch.float32, device=x.device)
            _run('window_attn_block_fwd', flops, lambda: L.check(L.lib.ococc_window_attn_block_train_fwd_bf16(
                L.ptr(x), L.ptr(pos), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, num_heads, L.ptr(wqkv), L.ptr(bq),
                L.ptr(wo), L.ptr(bo), L.ptr(g1), L.ptr(b1), float(eps), L.ptr(y), L.ptr(o_save), L.ptr(lse_save), L.stream()),
                'window_attn_block_train_fwd'))
        else:
            _run('window_attn_block_fwd', flops, lambda: L.check(L.lib.ococc_window_attn_block_fwd_bf16(
                L.ptr(x), L.ptr(pos), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, num_heads, L.ptr(wqkv), L.ptr(bq),
                L.ptr(wo), L.ptr(bo), L.ptr(g1), L.ptr(b1), float(eps), L.ptr(y), L.stream()), 'window_attn_block_fwd'))
        ctx.save_for_backward(x, pos, in_w, in_b, out_w, out_b, ln_w, *([o_save, lse_save] if keep else []))
        ctx.misc = (plan, float(eps), int(num_heads), bool(covered), wqkv, wo, bq, bo, g1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pos, in_w, in_b, out_w, out_b, ln_w = ctx.saved_tensors[:7]
        kept = ctx.saved_tensors[7:]
        plan, eps, H, covered, wqkv, wo, bq, bo, g1 = ctx.misc
        V, E = x.shape
        dy = dy.to(torch.bfloat16).contiguous()
        wot, wqkvt = linear_fragments([out_w.detach().t(), in_w.detach().t()])
        new = torch.empty if covered else torch.zeros
        dx = new((V, E), dtype=torch.bfloat16, device=x.device)
        dqkv = new((V, 3 * E), dtype=torch.bfloat16, device=x.device)
        dz = new((V, E), dtype=torch.bfloat16, device=x.device)
        prow = int(L.lib.ococc_window_block_partial_rows(plan.num_tiles))
        lnp = torch.empty((prow, 2, E), dtype=torch.float32, device=x.device)
        if kept:
            o, lse = kept
            L.check(L.lib.ococc_window_attn_block_bwd_saved_bf16(
                L.ptr(x), L.ptr(pos), L.ptr(dy), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, H, L.ptr(wqkv),
                L.ptr(bq), L.ptr(wo), L.ptr(bo), L.ptr(g1), eps, L.ptr(wot), L.ptr(wqkvt), L.ptr(o), L.ptr(lse), L.ptr(dx),
                L.ptr(dqkv), L.ptr(dz), L.ptr(lnp), L.stream()), 'window_attn_block_bwd_saved')
        else:
            o = new((V, E), dtype=torch.bfloat16, device=x.device)
            L.check(L.lib.ococc_window_attn_block_bwd_bf16(
                L.ptr(x), L.ptr(pos), L.ptr(dy), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, H, L.ptr(wqkv),
                L.ptr(bq), L.ptr(wo), L.ptr(bo), L.ptr(g1), eps, L.ptr(wot), L.ptr(wqkvt), L.ptr(dx), L.ptr(dqkv), L.ptr(dz),
                L.ptr(o), L.ptr(lnp), L.stream()), 'window_attn_block_bwd')
        # rows outside the plan hold zeros in dqkv / dz: they add nothing to the sums below
        (dwqkv, dbqkv), (dwo, dbo), (dg, db) = _wgrad([(dqkv, 3 * E, x, pos, 2 * E), (dz, E, o, None, 0)], V, x.device, lnp, prow)
        return (dx, None, None, dwqkv.to(in_w.dtype), dbqkv.to(in_b.dtype), dwo.to(out_w.dtype), dbo.to(out_b.dtype),
                dg.to(ln_w.dtype), db.to(ln_w.dtype), None, None, None)


class FfnBlock(torch.autograd.Function):
    """y2 = norm2(x + linear2(act(linear1(x)))) over all token rows."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ln_w, ln_b, eps, act):
        V, E = x.shape
        F = w1.shape[0]
        assert x.dtype == torch.bfloat16 and x.is_contiguous()
        f1, f2 = linear_fragments([w1.detach(), w2.detach()])
        c1, c2 = b1.detach().float().contiguous(), b2.detach().float().contiguous()
        g, b = ln_w.detach().float().contiguous(), ln_b.detach().float().contiguous()
        y = torch.empty_like(x)
        _run('token_ffn_block_fwd', 4.0 * V * E * F, lambda: L.check(L.lib.ococc_token_ffn_block_fwd_bf16(
            L.ptr(x), V, E, F, L.ptr(f1), L.ptr(c1), L.ptr(f2), L.ptr(c2), L.ptr(g), L.ptr(b), float(eps), ACT[act],
            L.ptr(y), L.stream()), 'token_ffn_block_fwd'))
        ctx.save_for_backward(x, w1, b1, w2, b2, ln_w)
        ctx.misc = (float(eps), act, f1, f2, c1, c2, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, ln_w = ctx.saved_tensors
        eps, act, f1, f2, c1, c2, g = ctx.misc
        V, E = x.shape
        F = w1.shape[0]
        dy = dy.to(torch.bfloat16).contiguous()
        f2t, f1t = linear_fragments([w2.detach().t(), w1.detach().t()])
        dev = x.device
        dx = torch.empty((V, E), dtype=torch.bfloat16, device=dev)
        a = torch.empty((V, F), dtype=torch.bfloat16, device=dev)
        dh = torch.empty((V, F), dtype=torch.bfloat16, device=dev)
        dz = torch.empty((V, E), dtype=torch.bfloat16, device=dev)
        prow = int(L.lib.ococc_window_block_partial_rows((V + TILE - 1) // TILE))
        lnp = torch.empty((prow, 2, E), dtype=torch.float32, device=dev)
        L.check(L.lib.ococc_token_ffn_block_bwd_bf16(
            L.ptr(x), L.ptr(dy), V, E, F, L.ptr(f1), L.ptr(c1), L.ptr(f2), L.ptr(c2), L.ptr(g), eps, ACT[act], L.ptr(f2t),
            L.ptr(f1t), L.ptr(dx), L.ptr(a), L.ptr(dh), L.ptr(dz), L.ptr(lnp), L.stream()), 'token_ffn_block_bwd')
        (dw1, db1), (dw2, db2), (dg, db) = _wgrad([(dh, F, x, None, 0), (dz, E, a, None, 0)], V, dev, lnp, prow)
        return (dx, dw1.to(w1.dtype), db1.to(b1.dtype), dw2.to(w2.dtype), db2.to(b2.dtype), dg.to(ln_w.dtype),
                db.to(ln_w.dtype), None, None)
