"""What a "rows with neighbours first" row order would buy the sub-manifold convolutions of configs[1].
45 % of the voxels of a uniformly random 3 % grid have no neighbour besides themselves; in the sorted row order they
are spread over every 16-row block.  Here the same voxels are fed in the order [rows with a neighbour | rows without]
(the general rulebook path takes unsorted coordinates) and the conv kernels are timed on both layouts.
Run on the GPU box: python tools/probe/perm_conv.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from objectcentricocccompletion_amd.spconv import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
B = 64
cells = torch.stack([torch.randperm(64000, generator=g)[:1970].sort().values + b * 64000 for b in range(B)]).flatten()
idx = torch.stack([cells // 64000, (cells // 1600) % 40, (cells // 40) % 40, cells % 40], 1).int().to(dev)
n = idx.shape[0]


def timed(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


_, pairs0, num0 = ops.get_indice_pairs(idx, B, [40, 40, 40], 3, subm=True)
table = pairs0._ococc.tables[(False, 'fwd')][0].view(27, n)
has = (table >= 0)
has[13] = False
lonely = ~has.any(0)
print('rows', n, 'without a neighbour', int(lonely.sum()), 'pairs per row', float(num0.sum()) / n)
perm = torch.cat([torch.nonzero(~lonely).squeeze(1), torch.nonzero(lonely).squeeze(1)])
layouts = {'sorted': idx, 'neighbours first': idx[perm].contiguous()}
for cin, cout in ((64, 128), (32, 64), (16, 32)):
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.05).to(dev)
    for name, ix in layouts.items():
        _, pairs, num = ops.get_indice_pairs(ix, B, [40, 40, 40], 3, subm=True)
        ops.set_rulebook_density(pairs, float(num.sum()) / n)
        x = torch.randn(n, cin, generator=g).to(dev).bfloat16()
        dy = torch.randn(n, cout, generator=g).to(dev).bfloat16()
        fwd = timed(lambda: ops.indice_conv(x, w, pairs, num, n, False, True))
        dgr = timed(lambda: ops.indice_conv_backward(x, w, dy, pairs, num, False, True, need_filter_grad=False))
        wgr = timed(lambda: ops.indice_conv_backward(x, w, dy, pairs, num, False, True, need_input_grad=False))
        print(f'{cin:4d} -> {cout:4d}  {name:17s} forward {fwd:6.1f} us   dgrad {dgr:6.1f} us   wgrad (+reduce) {wgr:6.1f} us')
