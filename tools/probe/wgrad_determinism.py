import sys, torch
sys.path.insert(0, '/root/repo')
from objectcentricocccompletion_amd.spconv import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
B = 8
cells = torch.stack([torch.randperm(64000, generator=g)[:1970].sort().values + b * 64000 for b in range(B)]).flatten()
idx = torch.stack([cells // 64000, (cells // 1600) % 40, (cells // 40) % 40, cells % 40], 1).int().to(dev)
n = idx.shape[0]
_, pairs, num = ops.get_indice_pairs(idx, B, [40, 40, 40], 3, subm=True)
for cin, cout in ((64, 128), (32, 64), (64, 64)):
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.05).to(dev)
    x = torch.randn(n, cin, generator=g).to(dev).bfloat16()
    dy = torch.randn(n, cout, generator=g).to(dev).bfloat16()
    outs = [ops.indice_conv_backward(x, w, dy, pairs, num, False, True, need_input_grad=False)[1].clone() for _ in range(6)]
    ref = torch.zeros(27, cin, cout, device=dev, dtype=torch.float64)
    tb = pairs._ococc.tables[(False, 'fwd')][0].view(27, n).long() if hasattr(pairs, '_ococc') and (False, 'fwd') in pairs._ococc.tables else None
    print(cin, cout, 'identical runs:', [bool(torch.equal(outs[0], o)) for o in outs[1:]], 'max', float(outs[0].abs().max()),
          'diff', max(float((outs[0] - o).abs().max()) for o in outs[1:]))
