"""Times the fused SST encoder-layer kernels one by one on synthetic tokens (run on the GPU box):
    python tools/probe/sst_block_bench.py [tokens] [mean window population]
260 k tokens in windows of ~10 tokens is the configs[4] per-GPU share.  OCOCC_WB_ABLATE=<mask> is passed through to the
backward kernels' debug knob (csrc/window_block.hip) to see what a phase costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from objectcentricocccompletion_amd import _lib as L  # noqa: E402
from objectcentricocccompletion_amd.sst import fused_block as fb  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 260000
    mean = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    lens = torch.poisson(torch.full((int(V / mean * 1.2),), mean), generator=g).clamp(1, 30).int()
    cs = torch.cumsum(lens, 0)
    nW = int((cs <= V).sum())
    lens = lens[:nW]
    V = int(lens.sum())
    T = 30
    tok = torch.full((nW * T,), -1, dtype=torch.int32)
    perm = torch.randperm(V, generator=g).int()
    start = torch.cumsum(lens, 0) - lens
    slot = torch.repeat_interleave(torch.arange(nW) * T, lens.long()) + (torch.arange(V) - torch.repeat_interleave(start, lens.long()))
    tok[slot] = perm
    plan = fb.TilePlan([(tok.to(dev), lens.to(dev), nW, T)], dev)
    print(f'{V} tokens, {nW} windows, {plan.num_tiles} tiles, fill {V / (plan.num_tiles * 64):.3f}')
    E, F, H = 128, 256, 8
    x = torch.randn(V, E, generator=g).bfloat16().to(dev)
    pos = torch.randn(V, E, generator=g).bfloat16().to(dev)
    dy = (torch.randn(V, E, generator=g) * 0.1).bfloat16().to(dev)
    P = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(dev).requires_grad_(True)
    in_w, in_b, out_w, out_b = P(3 * E, E), P(3 * E), P(E, E), P(E)
    w1, b1, w2, b2 = P(F, E), P(F), P(E, F), P(E)
    g1, be1, g2, be2 = (torch.ones(E, device=dev).requires_grad_(True) for _ in range(4))

    def timeit(name, fn, flops, n=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / n * 1e3
        print(f'{name:28s} {us:9.1f} us   {flops / us / 1e6:8.1f} TFLOP/s')

    wqkv, wo, f1, f2 = fb.linear_fragments([in_w.detach(), out_w.detach(), w1.detach(), w2.detach()])
    wot, wqkvt, f2t, f1t = fb.linear_fragments([out_w.detach().t(), in_w.detach().t(), w2.detach().t(), w1.detach().t()])
    bq, bo, c1, c2 = (t.detach().float().contiguous() for t in (in_b, out_b, b1, b2))
    gg1, bb1, gg2, bb2 = (t.detach().float().contiguous() for t in (g1, be1, g2, be2))
    y1 = torch.empty_like(x)
    y2 = torch.empty_like(x)
    attn_flops = V * 8.0 * E * E + 4.0 * E * plan.sum_sq
    timeit('attn_block_fwd', lambda: L.check(L.lib.ococc_window_attn_block_fwd_bf16(
        L.ptr(x), L.ptr(pos), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, H, L.ptr(wqkv), L.ptr(bq), L.ptr(wo),
        L.ptr(bo), L.ptr(gg1), L.ptr(bb1), 1e-5, L.ptr(y1), L.stream())), attn_flops)
    timeit('ffn_block_fwd', lambda: L.check(L.lib.ococc_token_ffn_block_fwd_bf16(
        L.ptr(y1), V, E, F, L.ptr(f1), L.ptr(c1), L.ptr(f2), L.ptr(c2), L.ptr(gg2), L.ptr(bb2), 1e-5, 0, L.ptr(y2),
        L.stream())), 4.0 * V * E * F)
    dx = torch.empty_like(x)
    a_ = torch.empty((V, F), dtype=torch.bfloat16, device=dev)
    dh = torch.empty_like(a_)
    dz = torch.empty_like(x)
    prow = int(L.lib.ococc_window_block_partial_rows((V + 63) // 64))
    lnp = torch.empty((prow, 2, E), dtype=torch.float32, device=dev)
    timeit('ffn_block_bwd', lambda: L.check(L.lib.ococc_token_ffn_block_bwd_bf16(
        L.ptr(y1), L.ptr(dy), V, E, F, L.ptr(f1), L.ptr(c1), L.ptr(f2), L.ptr(c2), L.ptr(gg2), 1e-5, 0, L.ptr(f2t),
        L.ptr(f1t), L.ptr(dx), L.ptr(a_), L.ptr(dh), L.ptr(dz), L.ptr(lnp), L.stream())), 12.0 * V * E * F)
    dqkv = torch.empty((V, 3 * E), dtype=torch.bfloat16, device=dev)
    o = torch.empty_like(x)
    prow2 = int(L.lib.ococc_window_block_partial_rows(plan.num_tiles))
    lnp2 = torch.empty((prow2, 2, E), dtype=torch.float32, device=dev)
    timeit('attn_block_bwd', lambda: L.check(L.lib.ococc_window_attn_block_bwd_bf16(
        L.ptr(x), L.ptr(pos), L.ptr(dy), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, H, L.ptr(wqkv), L.ptr(bq),
        L.ptr(wo), L.ptr(bo), L.ptr(gg1), 1e-5, L.ptr(wot), L.ptr(wqkvt), L.ptr(dx), L.ptr(dqkv), L.ptr(dz), L.ptr(o),
        L.ptr(lnp2), L.stream())), attn_flops + V * 8.0 * E * E + 8.0 * E * plan.sum_sq + attn_flops * 0)
    timeit('wgrad attn (qkv, o)', lambda: fb._wgrad([(dqkv, 3 * E, x, pos, 2 * E), (dz, E, o, None, 0)], V, dev), V * 8.0 * E * E)
    timeit('wgrad ffn (w1, w2)', lambda: fb._wgrad([(dh, F, y1, None, 0), (dz, E, a_, None, 0)], V, dev), V * 4.0 * E * F)


if __name__ == '__main__':
    main()
