"""The occupancy decoder's per-query MLP on csrc/mlp_layer.hip -- the MI355X form of OccDecoder.forward's conv_occ
(mmdet3d/models/occ/occ_base.py:99-153; build_mlp, mmdet3d/ops/sst/sst_ops.py:333-360):

    pe  = PosEncode(xyz)                                   ococc_pos_encode_bf16        bf16 [M, 64]
    y0  = drop(GELU(LN(pe W_pe^T + roi_part[idx])))        ococc_mlp_layer_fwd_bf16     (roi_part: once per RoI, f32)
    y1  = drop(GELU(LN(y0 W1^T)))                                  "
    out = drop(GELU(LN(y1 W2^T))) . w_head + b_head                "                     (y2 never leaves the chip)

One launch per layer, LayerNorm / GELU / dropout in the GEMM's epilogue; bf16 operands and activations, f32
accumulation and statistics."""
import ctypes
import os

import torch

from .. import _lib as L

_probe = None   # measurement hook (bench.py --workload decode): .wrap(name, flops, launch)


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


def pad16(k):
    return (k + 15) // 16 * 16


def pad64(k):
    return (k + 63) // 64 * 64


def linear_fragments32(mats, padded_cols=None):
    """f32 matrices [n, k] (2-D views, any strides, n a multiple of 32) -> bf16 fragment tensors of
    v_mfma_f32_32x32x16_bf16's A operand, columns zero-padded to ``padded_cols[i]``; one launch per 16."""
    padded_cols = list(padded_cols) if padded_cols is not None else [pad16(m.shape[1]) for m in mats]
    outs = []
    for i in range(0, len(mats), 16):
        part, pads = mats[i:i + 16], padded_cols[i:i + 16]
        for m, p in zip(part, pads):
            assert m.dim() == 2 and m.dtype == torch.float32 and m.shape[0] % 32 == 0 and p % 16 == 0 and p >= m.shape[1]
        dst = [torch.empty(m.shape[0] * p, dtype=torch.bfloat16, device=m.device) for m, p in zip(part, pads)]
        L.check(L.lib.ococc_linear_fragments32_bf16(
            len(part), _vp([m.data_ptr() for m in part]), _i64([m.shape[0] for m in part]), _i64([m.shape[1] for m in part]),
            _i64(pads), _i64([m.stride(0) for m in part]), _i64([m.stride(1) for m in part]),
            _vp([d.data_ptr() for d in dst]), L.stream()), 'linear_fragments32')
        outs += dst
    return outs


def pos_encode_bf16(xyz, num_freqs, bound=None, ld=None):
    """xyz f32 [M, 3] -> bf16 [M, ld] (ld = 6 L rounded up to 64): PosEncode.forward (occ_base.py:33-57), zero padded."""
    L.require_device(xyz)
    xyz = xyz.detach().float().contiguous()
    ld = pad64(6 * num_freqs) if ld is None else ld
    out = torch.empty((xyz.size(0), ld), dtype=torch.bfloat16, device=xyz.device)
    b = (ctypes.c_float * 6)(*[float(v) for v in bound]) if bound is not None else None
    L.check(L.lib.ococc_pos_encode_bf16(L.ptr(xyz), xyz.size(0), b, int(num_freqs), L.ptr(out), ld, L.stream()), 'pos_encode')
    return out


def mlp_layer(x, w_frag, n, ln_weight=None, ln_bias=None, eps=1e-5, act='gelu', bias=None, add_rows=None, add_index=None,
              drop_threshold=0, seed=0, head_weight=None, head_bias=None, want_y=True):
    """(y bf16 [M, n] or None, head f32 [M] or None) of one fused layer; x bf16 [M, k] contiguous."""
    L.require_device(x, w_frag)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()
    rows, k = x.shape
    assert w_frag.numel() == n * k, (w_frag.numel(), n, k)
    dev = x.device
    y = torch.empty((rows, n), dtype=torch.bfloat16, device=dev) if want_y else None
    head = torch.empty((rows,), dtype=torch.float32, device=dev) if head_weight is not None else None
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    bias, add_rows, ln_weight, ln_bias, head_weight, head_bias = map(
        f32, (bias, add_rows, ln_weight, ln_bias, head_weight, head_bias))
    if add_index is not None and add_index.dtype != torch.int32:
        add_index = add_index.to(torch.int32)
    code = {'none': 0, None: 0, 'gelu': 1}[act]
    flops = 2.0 * rows * n * k
    _run(f'mlp_layer_fwd_kernel k{k} n{n}', flops, lambda: L.check(L.lib.ococc_mlp_layer_fwd_bf16(
        L.ptr(x), rows, k, L.ptr(w_frag), n, L.ptr(bias), L.ptr(add_rows), L.ptr(add_index), L.ptr(ln_weight),
        L.ptr(ln_bias), float(eps), code, int(drop_threshold), int(seed), L.ptr(y), L.ptr(head_weight), L.ptr(head_bias),
        L.ptr(head), L.stream()), 'mlp_layer_fwd'))
    return y, head


OCC_MLP_WIDTHS = (64, 512, 1024, 1024)   # the one-launch kernel's shape: the reference's decoder (ococcnet.py occ_mlp)


def occ_mlp(pe, add_rows, add_index, w_frags, ln_weights, ln_biases, eps, head_weight, head_bias=None, drop_threshold=0,
            seeds=None, want_hidden=False):
    """All three layers + head in one launch (ococc_occ_mlp_fwd_bf16): logits f32 [M], and the hidden activations
    (y0 bf16 [M, 512], y1 bf16 [M, 1024]) when ``want_hidden``."""
    L.require_device(pe, add_rows)
    assert pe.dtype == torch.bfloat16 and pe.is_contiguous() and pe.shape[1] == OCC_MLP_WIDTHS[0]
    rows, dev = pe.size(0), pe.device
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    add_rows, head_weight, head_bias = f32(add_rows), f32(head_weight), f32(head_bias)
    gs, bs = [f32(t) for t in ln_weights], [f32(t) for t in ln_biases]
    assert add_rows.shape[1] == OCC_MLP_WIDTHS[1] and [g.numel() for g in gs] == list(OCC_MLP_WIDTHS[1:])
    assert [w.numel() for w in w_frags] == [OCC_MLP_WIDTHS[i + 1] * OCC_MLP_WIDTHS[i] for i in range(3)]
    if add_index.dtype != torch.int32:
        add_index = add_index.to(torch.int32)
    out = torch.empty((rows,), dtype=torch.float32, device=dev)
    y0 = torch.empty((rows, OCC_MLP_WIDTHS[1]), dtype=torch.bfloat16, device=dev) if want_hidden else None
    y1 = torch.empty((rows, OCC_MLP_WIDTHS[2]), dtype=torch.bfloat16, device=dev) if want_hidden else None
    sd = (ctypes.c_uint64 * 3)(*[int(s) for s in seeds]) if drop_threshold else None
    flops = 2.0 * rows * sum(OCC_MLP_WIDTHS[i + 1] * OCC_MLP_WIDTHS[i] for i in range(3))
    _run('occ_mlp_fwd_kernel', flops, lambda: L.check(L.lib.ococc_occ_mlp_fwd_bf16(
        L.ptr(pe), rows, L.ptr(add_rows), L.ptr(add_index), _vp([w.data_ptr() for w in w_frags]),
        _vp([g.data_ptr() for g in gs]), _vp([b.data_ptr() for b in bs]), float(eps), L.ptr(head_weight), L.ptr(head_bias),
        int(drop_threshold), sd, L.ptr(y0), L.ptr(y1), L.ptr(out), L.stream()), 'occ_mlp_fwd'))
    return (out, y0, y1) if want_hidden else out


class DecoderWeights(object):
    """bf16 operand fragments of the decoder's Linear weights, rebuilt when a parameter changes (its version counter or
    storage)."""

    def __init__(self):
        self._key = None
        self.frags = None

    def get(self, weights, pads):
        key = tuple((w.data_ptr(), w._version, tuple(w.shape), tuple(w.stride())) for w in weights)
        if key != self._key:
            self.frags = linear_fragments32([w.detach() for w in weights], pads)
            self._key = key
        return self.frags


# ---------------------------------------------------------------------------------------------------------------------
# Training: the forward is the same single launch (ococc_occ_mlp_train_fwd_bf16 also leaves z, the row statistics and
# y of every layer), the backward is the chain the separate operators ran -- LayerNorm backward kernels with the
# dropout masks regenerated from (threshold, seed), library GEMMs for dX and dW -- issued from one autograd node.
def wgrad_rows_bf16(dz, y, slices=32):
    """dz^T y -> f32 [n, k] for bf16 dz [M, n], y [M, k] with M in the 1e5..1e6: as ONE GEMM the library runs the
    [n, k] output as a few dozen macro tiles over a contraction of M (1024 x 1024 at 1 M rows: 4.1 ms, 0.54 PFLOP/s; 1024 x
    512: 3.4 ms) -- as a batched GEMM over row slices with f32 outputs, summed, 2.05 and 1.08 ms (tools/probe/
    dec_gemm_bench.py), and the partial sums are not rounded to bf16."""
    M, n = dz.shape
    per = M // slices
    if per < 512 or y.shape[1] < 16:
        return (dz.t() @ y).float()
    if n < 16:   # (the head's 1-wide gradient: as it is, the batched form takes a path that costs 11 ms of HOST time per call;
        wide = torch.zeros((M, 16), dtype=dz.dtype, device=dz.device)   # padded to 16 columns it is the fast one -- 1.4 ms
        wide[:, :n] = dz                                               # less per 64-tracklet step than one skinny GEMM)
        return wgrad_rows_bf16(wide, y, slices)[:n]
    main = per * slices
    out = torch.bmm(dz[:main].view(slices, per, dz.shape[1]).transpose(1, 2), y[:main].view(slices, per, y.shape[1]),
                    out_dtype=torch.float32).sum(0)
    if main < M:
        out = out + torch.mm(dz[main:].t(), y[main:], out_dtype=torch.float32)
    return out


# Three realisations of the backward pass (BACKWARD_MODE / OCOCC_DECODER_BACKWARD), the same numbers in all of them (z
# rounded to bf16 in front of every LayerNorm, bf16 d y between the layers, dropout masks from (threshold, seed)):
#   'chain' (default)  the forward leaves z / statistics / y of every layer row-major (10 KB per query row); the backward is
#                      the operator chain of _OccMlpTrain: own LayerNorm-backward kernels, library GEMMs for dX and dW.
#   'fused'            the forward parks z in the backward kernel's own lane order (+ statistics, y0, y1: 8 KB per row);
#                      the backward is ONE launch from d logit to d z0 (ococc_occ_mlp_bwd_bf16) + the weight gradients.
#   'recompute'        the forward keeps nothing but the logits; the one-launch backward runs the three layers forward
#                      again per 64-row tile.  Saves 10 KB of HBM per query row (10.7 GB at 64 tracklets).
# Measured on MI355X at 1 M query rows, dropout 0.1 (tools/probe/decoder_train_bench.py; tools/ab_decoder.sh inside the
# configs[2] step): forward + backward 18.7 ms chain / 19.9 ms fused / 23.3 ms recompute; step at 64 tracklets 56.5 / 61.1
# / 61.7 ms.  Why the one-launch forms lose although they move a fifth of the bytes: a workgroup that owns 64 rows and ALL
# channels streams the layer's whole weight matrix from L2 per 64 rows (the forward's design point, L2-bound at 0.28-0.36
# of the bf16 peak), while the library's 256 x 256 macro-tiles run the two input-gradient GEMMs at 1.3 PFLOP/s (0.52 of
# peak, 3.3 ms for both) -- and at 8 TB/s the activations' round trip through HBM that the fusion saves costs less than
# that difference; recompute adds the forward's 3.2 MFLOP per row on top.  The chain therefore stays the default; the
# one-launch kernel is the memory-saving option.  (EXPERIMENTS.md.)
RECOMPUTE_BACKWARD = True   # (kept for callers of round 5's first revision: False forces 'chain')
BACKWARD_MODE = os.environ.get('OCOCC_DECODER_BACKWARD', 'chain')


class _OccMlpTrainRecompute(torch.autograd.Function):
    """Same contract and the same numbers as _OccMlpTrain (z rounded to bf16 in front of every LayerNorm, bf16 operands,
    f32 accumulation, dropout masks from (threshold, seed)); the backward is ONE launch (+ the weight-gradient
    contractions), in BACKWARD_MODE 'fused' on what the forward parked, in 'recompute' on nothing but the inputs."""

    @staticmethod
    def forward(ctx, pe, roi_part, idx, w_pe, w1, w2, g0, b0, g1, b1, g2, b2, head_w, head_b, eps, drop_threshold, seeds,
                cache):
        M, dev = pe.size(0), pe.device
        frags = cache.get([w_pe, w1, w2], [pe.shape[1], w1.shape[1], w2.shape[1]])
        f32 = lambda t: t.detach().float().contiguous()
        gs, bs = [f32(g0), f32(g1), f32(g2)], [f32(b0), f32(b1), f32(b2)]
        add = f32(roi_part)
        hw, hb = f32(head_w).view(-1), f32(head_b).view(-1)
        out = torch.empty((M,), dtype=torch.float32, device=dev)
        sd = (ctypes.c_uint64 * 3)(*[int(v) for v in seeds]) if drop_threshold else None
        flops = 2.0 * M * sum(OCC_MLP_WIDTHS[i + 1] * OCC_MLP_WIDTHS[i] for i in range(3))
        recompute = BACKWARD_MODE == 'recompute'
        kept = []
        if recompute:
            zv = yv = sv = None
        else:   # z parked in whole 64-row tiles, the statistics, y0 / y1 (the weight gradients' operands)
            Mt = (M + 63) // 64 * 64
            zs = [torch.empty((Mt, n), dtype=torch.bfloat16, device=dev) for n in OCC_MLP_WIDTHS[1:]]
            ys = [torch.empty((M, n), dtype=torch.bfloat16, device=dev) for n in OCC_MLP_WIDTHS[1:3]]
            stats = [torch.empty((M, 2), dtype=torch.float32, device=dev) for _ in range(3)]
            zv, yv, sv = _vp([z.data_ptr() for z in zs]), _vp([y.data_ptr() for y in ys] + [0]), _vp([t.data_ptr() for t in stats])
            kept = zs + ys + stats
        _run('occ_mlp_fwd_kernel (training)', flops, lambda: L.check(L.lib.ococc_occ_mlp_train_fwd_bf16(
            L.ptr(pe), M, L.ptr(add), L.ptr(idx), _vp([w.data_ptr() for w in frags]), _vp([g.data_ptr() for g in gs]),
            _vp([b.data_ptr() for b in bs]), float(eps), L.ptr(hw), L.ptr(hb), int(drop_threshold), sd,
            zv, yv, sv, L.ptr(out), L.stream()), 'occ_mlp_train_fwd'))
        ctx.save_for_backward(pe, idx, add, w_pe, w1, w2, head_w, *gs, *bs, *kept)
        ctx.ln_params = ((g0, b0), (g1, b1), (g2, b2))
        ctx.misc = (float(eps), int(drop_threshold), tuple(int(v) for v in seeds), roi_part.size(0), cache, frags, recompute)
        return out.view(M, 1)

    @staticmethod
    def backward(ctx, dlogit):
        from ..linear import sliced_wgrad
        t = ctx.saved_tensors
        pe, idx, add, w_pe, w1, w2, head_w = t[:7]
        gs, bs = t[7:10], t[10:13]
        eps, thr, seeds, K, cache, frags, recompute = ctx.misc
        M, dev = pe.size(0), pe.device
        bf = torch.bfloat16
        if not hasattr(cache, 'transposed'):
            cache.transposed = DecoderWeights()
        frags = cache.get([w_pe, w1, w2], [pe.shape[1], w1.shape[1], w2.shape[1]])   # (unchanged parameters: the forward's)
        frags_t = cache.transposed.get([w1.t(), w2.t()], [w1.shape[0], w2.shape[0]])
        n0, n1, n2 = OCC_MLP_WIDTHS[1:]
        if recompute:
            y0, y1 = torch.empty((M, n0), dtype=bf, device=dev), torch.empty((M, n1), dtype=bf, device=dev)
            scratch = L.workspace(int(L.lib.ococc_occ_mlp_bwd_scratch_bytes(M)), dev)
            zv = sv = None
        else:
            zs, (y0, y1), stats = t[13:16], t[16:18], t[18:21]
            scratch = None
            zv, sv = _vp([z.data_ptr() for z in zs]), _vp([x.data_ptr() for x in stats])
        dz0, dz1, dz2 = (torch.empty((M, n), dtype=bf, device=dev) for n in (n0, n1, n2))
        wgs, cols = int(L.lib.ococc_occ_mlp_bwd_workgroups(M)), int(L.lib.ococc_occ_mlp_bwd_partial_cols())
        partials = torch.zeros((wgs, cols), dtype=torch.float32, device=dev)
        d = dlogit.reshape(M).float().contiguous()
        hw = head_w.detach().float().contiguous().view(-1)
        sd = (ctypes.c_uint64 * 3)(*[int(v) for v in seeds]) if thr else None
        flops = 2.0 * M * ((sum(OCC_MLP_WIDTHS[i + 1] * OCC_MLP_WIDTHS[i] for i in range(3)) if recompute else 0) + n2 * n1 + n1 * n0)
        _run('occ_mlp_bwd_kernel' + (' (recompute)' if recompute else ''), flops, lambda: L.check(L.lib.ococc_occ_mlp_bwd_bf16(
            L.ptr(pe), M, L.ptr(add), L.ptr(idx), _vp([w.data_ptr() for w in frags]), _vp([w.data_ptr() for w in frags_t] + [0]),
            _vp([g.data_ptr() for g in gs]), _vp([b.data_ptr() for b in bs]), float(eps), L.ptr(hw), L.ptr(d), int(thr), sd,
            zv, sv, _vp([y0.data_ptr(), y1.data_ptr(), 0]), _vp([dz0.data_ptr(), dz1.data_ptr(), dz2.data_ptr()]),
            L.ptr(partials), L.ptr(scratch), scratch.numel() if scratch is not None else 0, L.stream()), 'occ_mlp_bwd'))
        sums = partials.sum(0)
        off, grads_ln = 0, []
        for l, n in enumerate((n0, n1, n2)):
            lw, lb = ctx.ln_params[l]
            grads_ln += [sums[off:off + n].to(lw.dtype), sums[off + n:off + 2 * n].to(lb.dtype)]
            off += 2 * n
        d_head_w = sums[off:off + n2].view(1, n2).to(head_w.dtype)
        d_head_b = dlogit.sum().reshape(1)
        # weight gradients: contractions over the rows of what the kernel left
        dw2 = wgrad_rows_bf16(dz2, y1).to(w2.dtype)
        dw1 = wgrad_rows_bf16(dz1, y0).to(w1.dtype)
        dz0f = dz0.float()
        dw0 = sliced_wgrad(dz0f, pe[:, :w_pe.shape[1]].float()).to(w_pe.dtype)
        d_roi = torch.empty((K, n0), dtype=torch.float32, device=dev)
        L.check(L.lib.ococc_segment_reduce_f32(L.ptr(dz0f), L.ptr(idx), M, n0, 0, None, L.ptr(d_roi), None, K, L.stream()),
                'occ_mlp_train_bwd: roi_part')
        return (None, d_roi, None, dw0, dw1, dw2, grads_ln[0], grads_ln[1], grads_ln[2], grads_ln[3], grads_ln[4],
                grads_ln[5], d_head_w, d_head_b, None, None, None, None)


class _OccMlpTrain(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pe, roi_part, idx, w_pe, w1, w2, g0, b0, g1, b1, g2, b2, head_w, head_b, eps, drop_threshold, seeds,
                cache):
        """pe bf16 [M, 64] (no gradient), roi_part f32 [K, 512], idx int32 [M] non-decreasing, w_pe [512, pe columns],
        w1 [1024, 512], w2 [1024, 1024] f32 parameters (views allowed), LayerNorm parameters, head_w [1, 1024], head_b [1]."""
        from ..norm import layernorm_act_backward  # noqa: F401 (imported here: norm imports nothing from this module)
        M, dev = pe.size(0), pe.device
        frags = cache.get([w_pe, w1, w2], [pe.shape[1], w1.shape[1], w2.shape[1]])
        f32 = lambda t: t.detach().float().contiguous()
        gs, bs = [f32(g0), f32(g1), f32(g2)], [f32(b0), f32(b1), f32(b2)]
        add = f32(roi_part)
        hw, hb = f32(head_w).view(-1), f32(head_b).view(-1)
        widths = OCC_MLP_WIDTHS[1:]
        zs = [torch.empty((M, n), dtype=torch.bfloat16, device=dev) for n in widths]
        ys = [torch.empty((M, n), dtype=torch.bfloat16, device=dev) for n in widths]
        stats = [torch.empty((M, 2), dtype=torch.float32, device=dev) for _ in widths]
        out = torch.empty((M,), dtype=torch.float32, device=dev)
        sd = (ctypes.c_uint64 * 3)(*[int(v) for v in seeds]) if drop_threshold else None
        flops = 2.0 * M * sum(OCC_MLP_WIDTHS[i + 1] * OCC_MLP_WIDTHS[i] for i in range(3))
        _run('occ_mlp_fwd_kernel (training)', flops, lambda: L.check(L.lib.ococc_occ_mlp_train_fwd_bf16(
            L.ptr(pe), M, L.ptr(add), L.ptr(idx), _vp([w.data_ptr() for w in frags]), _vp([g.data_ptr() for g in gs]),
            _vp([b.data_ptr() for b in bs]), float(eps), L.ptr(hw), L.ptr(hb), int(drop_threshold), sd,
            _vp([z.data_ptr() for z in zs]), _vp([y.data_ptr() for y in ys]), _vp([t.data_ptr() for t in stats]), L.ptr(out),
            L.stream()), 'occ_mlp_train_fwd'))
        ctx.save_for_backward(pe, idx, w_pe, w1, w2, head_w, *gs, *bs, *zs, *ys, *stats)
        ctx.ln_params = ((g0, b0), (g1, b1), (g2, b2))
        ctx.misc = (float(eps), int(drop_threshold), tuple(int(v) for v in seeds), roi_part.size(0))
        return out.view(M, 1)

    @staticmethod
    def backward(ctx, dlogit):
        from ..linear import sliced_wgrad
        from ..norm import layernorm_act_backward
        t = ctx.saved_tensors
        pe, idx, w_pe, w1, w2, head_w = t[:6]
        gs, bs, zs, ys, stats = t[6:9], t[9:12], t[12:15], t[15:18], t[18:21]
        eps, thr, seeds, K = ctx.misc
        M, dev = pe.size(0), pe.device
        bf = torch.bfloat16
        d = dlogit.reshape(M, 1).to(bf)
        # head: logit = y2 . w_head + b_head
        d_head_w = wgrad_rows_bf16(d, ys[2]).to(head_w.dtype)
        d_head_b = dlogit.sum().reshape(1)
        dy = d @ head_w.detach().to(bf).view(1, -1)                                  # [M, 1024]
        weights = (w_pe, w1, w2)
        grads_ln, dws = [None] * 6, [None] * 3
        dz0 = None
        for l in (2, 1, 0):
            dz = torch.empty_like(zs[l])
            lw, lb = ctx.ln_params[l]
            dg, db = layernorm_act_backward(zs[l], dy.contiguous(), gs[l], bs[l], stats[l], 1, dz, lw, lb,
                                            drop=(thr, seeds[l]) if thr else (0, 0))
            grads_ln[2 * l], grads_ln[2 * l + 1] = (None if dg is None else dg.to(lw.dtype)), (None if db is None else db.to(lb.dtype))
            if l > 0:
                dws[l] = wgrad_rows_bf16(dz, ys[l - 1]).to(weights[l].dtype)         # [n_l, n_{l-1}]
                dy = dz @ weights[l].detach().to(bf)                                 # [M, n_{l-1}]
            else:
                dz0 = dz.float()
        # first layer: z0 = pe W_pe^T + roi_part[idx]
        pe32 = pe[:, :w_pe.shape[1]].float()
        dws[0] = sliced_wgrad(dz0, pe32).to(w_pe.dtype)
        d_roi = torch.empty((K, dz0.shape[1]), dtype=torch.float32, device=dev)
        L.check(L.lib.ococc_segment_reduce_f32(L.ptr(dz0), L.ptr(idx), M, dz0.shape[1], 0, None, L.ptr(d_roi), None, K,
                                               L.stream()), 'occ_mlp_train_bwd: roi_part')
        return (None, d_roi, None, dws[0], dws[1], dws[2], grads_ln[0], grads_ln[1], grads_ln[2], grads_ln[3], grads_ln[4],
                grads_ln[5], d_head_w, d_head_b, None, None, None, None)


def occ_mlp_train(pe, roi_part, idx, w_pe, w1, w2, ln_weights, ln_biases, eps, head_w, head_b, drop_threshold, seeds, cache):
    """logits f32 [M, 1] with a backward pass; see _OccMlpTrain."""
    assert pe.dtype == torch.bfloat16 and pe.shape[1] == OCC_MLP_WIDTHS[0] and idx.dtype == torch.int32
    fn = _OccMlpTrain if (BACKWARD_MODE == 'chain' or not RECOMPUTE_BACKWARD) else _OccMlpTrainRecompute
    return fn.apply(pe, roi_part, idx, w_pe, w1, w2, ln_weights[0], ln_biases[0], ln_weights[1], ln_biases[1],
                    ln_weights[2], ln_biases[2], head_w, head_b, eps, drop_threshold, seeds, cache)
