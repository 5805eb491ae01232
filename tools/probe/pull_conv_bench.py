"""64 -> 128 sub-manifold forward at the benchmark size (64 grids of 40^3, 2000 random points each): the pull kernel
against the streamed-weights kernel.  usage: python tools/probe/pull_conv_bench.py [cin cout]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from objectcentricocccompletion_amd.spconv import ops  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    B, S, P = 64, 40, 2000
    cells = torch.stack([torch.randperm(S ** 3, generator=g)[:P] for _ in range(B)])   # distinct cells per grid
    cells, _ = cells.sort(1)
    b = torch.arange(B).view(-1, 1).expand(-1, P)
    idx = torch.stack([b, cells // (S * S), (cells // S) % S, cells % S], -1).view(-1, 4).int().to(dev)
    _, pairs, num = ops.get_indice_pairs(idx, B, [S, S, S], 3, subm=True)
    n = idx.size(0)
    ppr = float(num.sum().item()) / n
    x = torch.randn(n, cin, device=dev).bfloat16()
    w = (torch.randn(3, 3, 3, cin, cout, device=dev) * 0.1)
    print(f'{n} rows, {ppr:.2f} pairs per row, {cin} -> {cout}')
    ops.set_rulebook_density(pairs, ppr)
    res = {}
    for name, pull in (('stream / tile (as chosen today)', False), ('pull', True)):
        ops.PULL_CONV = pull
        try:
            y = ops.indice_conv(x, w, pairs, num, n, False, True)
            t = timed(lambda: ops.indice_conv(x, w, pairs, num, n, False, True))
        finally:
            ops.PULL_CONV = None
        res[name] = y
        alg = n * cin * 2 + n * cout * 2 + float(num.sum().item()) * 8 + 27 * cin * cout * 2
        print(f'{name:34s} {t:7.1f} us  ({alg / t / 1e6:.2f} TB/s algorithmic, {alg / t / 1e6 / 8:.3f} of 8 TB/s)', flush=True)
    import ctypes
    from objectcentricocccompletion_amd import _lib as L
    raw = ctypes.CDLL(L.LIB_PATH)
    ops.PULL_CONV = True
    for rows in (256, 512):
        raw.ococc_sparse_conv_pull_probe(rows << 8)
        y = ops.indice_conv(x, w, pairs, num, n, False, True)
        t = timed(lambda: ops.indice_conv(x, w, pairs, num, n, False, True))
        print(f'   pull, {rows}-row tiles: {t:7.1f} us   max abs diff vs stream {float((y.float() - res["stream / tile (as chosen today)"].float()).abs().max())}', flush=True)
    for mask, what in ((1, 'no sparse rounds'), (2, 'no dense pass'), (3, 'lists + stores only'), (7, 'lists only'), (4, 'no stores')):
        raw.ococc_sparse_conv_pull_probe(mask)
        t = timed(lambda: ops.indice_conv(x, w, pairs, num, n, False, True))
        print(f'   pull, {what:22s} {t:7.1f} us', flush=True)
    raw.ococc_sparse_conv_pull_probe(0)
    ops.PULL_CONV = None
    a, b_ = list(res.values())
    print('max abs difference', float((a.float() - b_.float()).abs().max()))


if __name__ == '__main__':
    main()
