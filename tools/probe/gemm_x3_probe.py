#!/usr/bin/env python3
"""f32 library GEMM against ONE bf16 GEMM over the three-way split operands (x = hi + lo, products hi hi + hi lo + lo hi,
f32 accumulation: K' = 3 K) for the token-stack / head shapes of configs[2], per row count M.  Prints time and error vs f64."""
import sys, time
import torch

dev = torch.device('cuda:0')
shapes = [(3072, 1536), (1536, 1536), (512, 1536), (1536, 512), (2048, 3072), (2048, 2048), (1536, 2048), (512, 512)]


def split(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi, lo


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (128, 256, 512, 1024, 2048):
    for N, K in shapes:
        g = torch.Generator(device=dev).manual_seed(N + K + M)
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
        xh, xl = split(x)
        wh, wl = split(w)
        x3 = torch.cat([xh, xh, xl], 1).contiguous()
        w3 = torch.cat([wh, wl, wh], 1).contiguous()
        t32 = bench(lambda: torch.mm(x, w.t()))
        t3 = bench(lambda: torch.mm(x3, w3.t(), out_dtype=torch.float32))
        t1 = bench(lambda: torch.mm(xh, wh.t(), out_dtype=torch.float32))
        ref = x.double() @ w.double().t()
        e32 = float((torch.mm(x, w.t()).double() - ref).norm() / ref.norm())
        e3 = float((torch.mm(x3, w3.t(), out_dtype=torch.float32).double() - ref).norm() / ref.norm())
        e1 = float((torch.mm(xh, wh.t(), out_dtype=torch.float32).double() - ref).norm() / ref.norm())
        print(f'M {M:5d} N {N:5d} K {K:5d}: f32 {t32:7.1f} us  bf16x3 {t3:7.1f} us  bf16 {t1:7.1f} us   err f32 {e32:.1e} x3 {e3:.1e} bf16 {e1:.1e}', flush=True)
