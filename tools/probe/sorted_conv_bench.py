"""64 -> 128 (and other) sub-manifold layers on the configs[1] scene: neighbour-pattern row order against voxel order.
usage: python tools/probe/sorted_conv_bench.py [cin cout]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from objectcentricocccompletion_amd.spconv import ops  # noqa: E402

dev = torch.device('cuda:0')
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
only = sys.argv[3] if len(sys.argv) > 3 else None   # e.g. "4,8": one sorted configuration (for a kernel trace)
g = torch.Generator().manual_seed(0)
G, S, P = 64, 40, 2000
cells = []
for b in range(G):
    p = torch.randint(0, S, (P, 3), generator=g)
    flat = torch.unique(p[:, 0] * S * S + p[:, 1] * S + p[:, 2])
    cells.append(torch.stack([torch.full_like(flat, b), flat // (S * S), (flat // S) % S, flat % S], 1))
coors = torch.cat(cells).to(torch.int32).to(dev)
n = coors.shape[0]
w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.1).to(dev)
x = torch.randn(n, cin, generator=g).to(dev).bfloat16()
print('rows', n)


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps // 10):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps // 10 * 10)


ops.SPARSE_TILE_CONV = False
ref = None
for label, on, tiles in (('voxel order', False, (4, 8)), ('sorted 4,8', True, (4, 8)), ('sorted 4,4', True, (4, 4)),
                         ('sorted 8,8', True, (8, 8)), ('sorted 4,16', True, (4, 16)), ('sorted 16,16', True, (16, 16))):
    if only is not None:
        if not on:
            continue
        v = [int(t) for t in only.split(',')]
        label, tiles = 'sorted ' + only, (v[0], v[1])
    ops.SORTED_CONV, ops.SORTED_TILES = on, tiles
    _, pairs, num = ops.get_indice_pairs(coors, G, [S, S, S], 3, subm=True)
    y = ops.indice_conv(x, w, pairs, num, n, False, True)
    if ref is None:
        ref = y
    torch.cuda.synchronize()
    us = timed(lambda: ops.indice_conv(x, w, pairs, num, n, False, True))
    line = f'{label:20s} {us:7.1f} us   equal to voxel order: {bool(torch.equal(ref, y))}'
    if on:
        rb, (table, mask, rows) = ops._tables_for(pairs, num, False, 'fwd', n, True)
        rb.orders.clear()
        t_order = timed(lambda: (rb.orders.clear(), ops.row_order(rb, table, rows)))
        line += f'   building the order: {t_order:6.1f} us   hdr {ops.row_order(rb, table, rows)[1].tolist()[:4]}'
    print(line)
    if only is not None:
        break
