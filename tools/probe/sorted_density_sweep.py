"""Neighbour-pattern row order against voxel order across neighbourhood densities (random cells per 40^3 grid, 64 grids),
64 -> 128 forward; microseconds per call, 10 launches per graph replay.  Basis of spconv.ops.SORTED_MAX_PAIRS_PER_ROW."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from objectcentricocccompletion_amd.spconv import ops

dev = torch.device('cuda:0')
B, cin, cout = 64, 64, 128
ops.SPARSE_TILE_CONV = False
for vox in (500, 1000, 2000, 3000, 4000, 6000, 8000, 12000):
    g = torch.Generator().manual_seed(3)
    cells = torch.stack([torch.randperm(64000, generator=g)[:vox].sort().values + b * 64000 for b in range(B)]).flatten()
    idx = torch.stack([cells // 64000, (cells // 1600) % 40, (cells // 40) % 40, cells % 40], 1).int().to(dev)
    n = idx.shape[0]
    x = torch.randn(n, cin, generator=g).to(dev).bfloat16()
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.05).to(dev)
    out = []
    for srt in (False, True):
        ops.SORTED_CONV = srt
        _, pairs, num = ops.get_indice_pairs(idx, B, [40, 40, 40], 3, subm=True)
        ppr = float(num.sum()) / n
        run = lambda: ops.indice_conv(x, w, pairs, num, n, False, True)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                run()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 10)
    print(f'rows {n:7d} pairs/row {ppr:5.2f}   voxel order {out[0]:7.1f} us   pattern order {out[1]:7.1f} us (+ ~18 us to build the order)', flush=True)
