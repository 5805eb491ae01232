"""Micro driver for profiling one sparse-conv shape (used with rocprofv3 --pmc / --kernel-trace)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objectcentricocccompletion_amd.spconv import ops

ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=64)
ap.add_argument('--cout', type=int, default=128)
ap.add_argument('--grids', type=int, default=64)
ap.add_argument('--vox', type=int, default=2000)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--mode', default='fwd', choices=['fwd', 'bwd', 'both'])
ap.add_argument('--tile', action='store_true', help='force the compact-then-multiply kernel (ococc_sparse_conv_tile_bf16)')
ap.add_argument('--sorted', action='store_true', help='force the neighbour-pattern row order (ococc_sparse_conv_sorted_bf16)')
a = ap.parse_args()
dev = torch.device('cuda:0')
if a.tile:
    ops.SPARSE_TILE_CONV = True
if a.sorted:
    ops.SPARSE_TILE_CONV, ops.SORTED_CONV = False, True
g = torch.Generator().manual_seed(3)
B = a.grids
cells = torch.stack([torch.randperm(64000, generator=g)[:a.vox].sort().values + b * 64000 for b in range(B)]).flatten()
idx = torch.stack([cells // 64000, (cells // 1600) % 40, (cells // 40) % 40, cells % 40], 1).int().to(dev)
n = idx.shape[0]
_, pairs, num = ops.get_indice_pairs(idx, B, [40, 40, 40], 3, subm=True)
x = torch.randn(n, a.cin, generator=g).to(dev).bfloat16()
dy = torch.randn(n, a.cout, generator=g).to(dev).bfloat16()
w = (torch.randn(3, 3, 3, a.cin, a.cout, generator=g) * 0.05).to(dev)
torch.cuda.synchronize()
for _ in range(a.iters):
    if a.mode in ('fwd', 'both'):
        y = ops.indice_conv(x, w, pairs, num, n, False, True)
    if a.mode in ('bwd', 'both'):
        din, dw = ops.indice_conv_backward(x, w, dy, pairs, num, False, True)
torch.cuda.synchronize()
print('ok', n, int(num.sum()))
