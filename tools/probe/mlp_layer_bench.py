"""Per-layer rate of the fused decoder-layer kernel (csrc/mlp_layer.hip) at ~1 M query rows, next to the library
GEMM + LayerNorm kernel pair it replaces.  usage: python tools/probe/mlp_layer_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from objectcentricocccompletion_amd.occ import fused_mlp as fm  # noqa: E402
from objectcentricocccompletion_amd.norm import layer_norm_act  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    whole_only = len(sys.argv) > 2 and sys.argv[2] == 'whole'
    for k, n, head in () if whole_only else ((64, 512, False), (512, 1024, False), (1024, 1024, False), (1024, 1024, True)):
        x = torch.randn(rows, k, device=dev).to(torch.bfloat16)
        W = torch.randn(n, k, device=dev) / k ** 0.5
        gam, bet = torch.ones(n, device=dev), torch.zeros(n, device=dev)
        hw, hb = torch.randn(n, device=dev), torch.zeros(1, device=dev)
        wf, = fm.linear_fragments32([W], [k])
        add = torch.randn(2048, n, device=dev) if k == 64 else None
        idx = torch.arange(rows, device=dev, dtype=torch.int32) // (rows // 2048 + 1) if k == 64 else None
        fused = lambda: fm.mlp_layer(x, wf, n, gam, bet, 1e-3, 'gelu', add_rows=add, add_index=idx,
                                     head_weight=hw if head else None, head_bias=hb if head else None, want_y=not head)
        Wb = W.to(torch.bfloat16)
        lib = lambda: layer_norm_act(x @ Wb.t(), gam, bet, 1e-3, 'gelu')
        tf, tl = timed(fused), timed(lib)
        fl = 2.0 * rows * n * k
        if head:
            bare = lambda: fm.mlp_layer(x, wf, n, None, None, 0.0, 'none', head_weight=hw, head_bias=hb, want_y=False)
            tb = timed(bare)
            print(f'   (GEMM + head dot only, no LayerNorm / GELU: {tb:7.3f} ms {2.0 * rows * n * k / tb / 1e9:7.1f} TF/s)')
        print(f'k={k:5d} n={n:5d} head={int(head)} rows={rows}: fused {tf:7.3f} ms {fl / tf / 1e9:7.1f} TF/s | '
              f'library GEMM + LN kernel {tl:7.3f} ms {fl / tl / 1e9:7.1f} TF/s', flush=True)
    # the whole MLP in one launch
    W = [torch.randn(n, k, device=dev) / k ** 0.5 for k, n in ((64, 512), (512, 1024), (1024, 1024))]
    frags = fm.linear_fragments32(W, [64, 512, 1024])
    gam = [torch.ones(n, device=dev) for n in (512, 1024, 1024)]
    bet = [torch.zeros(n, device=dev) for n in (512, 1024, 1024)]
    pe = torch.randn(rows, 64, device=dev).to(torch.bfloat16)
    add = torch.randn(2048, 512, device=dev)
    idx = torch.arange(rows, device=dev, dtype=torch.int32) // (rows // 2048 + 1)
    hw, hb = torch.randn(1024, device=dev), torch.zeros(1, device=dev)
    fl = 2.0 * rows * (64 * 512 + 512 * 1024 + 1024 * 1024)
    for hidden in (False, True):
        t = timed(lambda: fm.occ_mlp(pe, add, idx, frags, gam, bet, 1e-3, hw, hb, want_hidden=hidden))
        print(f'whole MLP, one launch, hidden activations {"stored" if hidden else "on chip"}: {t:7.3f} ms '
              f'{fl / t / 1e9:7.1f} TF/s', flush=True)
    # one launch at a time with the device idle in between (how a decode step meets the kernel)
    import time
    for gap in (0.0, 0.002, 0.02, 0.2):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            time.sleep(gap)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fm.occ_mlp(pe, add, idx, frags, gam, bet, 1e-3, hw, hb)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f'single launches, {gap * 1e3:5.1f} ms idle before each: ' + ' '.join(f'{t:6.3f}' for t in ts) + ' ms', flush=True)


if __name__ == '__main__':
    main()
