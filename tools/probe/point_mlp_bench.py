"""Times the fused per-point SIR layer kernels (csrc/point_mlp.hip) at the configs[2] B = 64 size (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from objectcentricocccompletion_amd.point_mlp import point_layer  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    G = M // 64
    inv = torch.repeat_interleave(torch.arange(G), 64).int().to(dev)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)

    def timeit(name, fn, flops, n=10):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / n * 1e3
        print(f'{name:40s} {us:9.1f} us   {flops / us / 1e6:8.1f} TFLOP/s')

    for name, ka, kmul, kb, kv, n, mx in (('rel 13->16', 13, 0, 0, 0, 16, False), ('rel 16->32', 16, 0, 0, 0, 32, False),
                                          ('rel 32->144', 32, 0, 0, 0, 144, False), ('vfe0 144->128 (gate, max)', 144, 1, 0, 0, 128, True),
                                          ('vfe0 24->128 (gate, max)', 24, 1, 0, 0, 128, True),
                                          ('vfe1 128+128->128 (gather, max)', 128, 0, 0, 128, 128, True)):
        a = R(M, ka).requires_grad_(True)
        mul = R(M, ka).requires_grad_(True) if kmul else None
        v = R(G, kv).requires_grad_(True) if kv else None
        w = (R(n, ka + kb + kv) / (ka + kv) ** 0.5).requires_grad_(True)
        gam, bet = torch.ones(n, device=dev, requires_grad=True), torch.zeros(n, device=dev, requires_grad=True)
        dy, dm = R(M, n), R(G, n)
        kw = dict(mul=mul, v=v, inv=inv if (kv or mx) else None, num_segments=G, seg_max=mx)
        flops = 2.0 * M * n * (ka + kb + kv)

        def fwd():
            with torch.no_grad():
                point_layer(a, w, gam, bet, 1e-3, 'gelu', **kw)

        def fwd_bwd():
            for t in (a, mul, v, w, gam, bet):
                if t is not None:
                    t.grad = None
            out = point_layer(a, w, gam, bet, 1e-3, 'gelu', **kw)
            if mx:
                (out[0] * dy).sum().backward(retain_graph=False, inputs=None) if False else torch.autograd.backward(
                    [out[0], out[1]], [dy, dm])
            else:
                out.backward(dy)
        timeit(name + ' fwd', fwd, flops)
        timeit(name + ' fwd+bwd', fwd_bwd, 4 * flops)


if __name__ == '__main__':
    main()
