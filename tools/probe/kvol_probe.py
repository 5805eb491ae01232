import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import oracle as O
from objectcentricocccompletion_amd.spconv import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
for ks in ((5, 5, 5), (1, 7, 7), (3, 3, 5)):
    B, shape = 2, (12, 13, 14)
    cells = np.stack(np.meshgrid(np.arange(B), *[np.arange(s) for s in shape], indexing='ij'), -1).reshape(-1, 4)
    idx = cells[rng.random(len(cells)) < 0.2].astype(np.int32)
    n = len(idx)
    cin, cout = 16, 32
    x = O.bf16_round(rng.standard_normal((n, cin)).astype(np.float32))
    w = O.bf16_round(rng.standard_normal(ks + (cin, cout)).astype(np.float32) * 0.2)
    dy = O.bf16_round(rng.standard_normal((n, cout)).astype(np.float32))
    try:
        _, pairs, num = ops.get_indice_pairs(torch.from_numpy(idx).to(dev), B, list(shape), list(ks), subm=True)
        ep, en = O.subm_rulebook(idx, B, shape, ksize=ks)
        print(ks, 'rulebook num equal', np.array_equal(num.cpu().numpy(), en))
        xt, wt, dyt = (torch.from_numpy(a).to(dev) for a in (x, w, dy))
        y = ops.indice_conv(xt, wt, pairs, num, n, False, True)
        ey = O.indice_conv(x, w, ep, en, n, subm=True)
        print(ks, 'fwd max err', float(np.abs(y.cpu().numpy() - ey).max()))
        din, dw = ops.indice_conv_backward(xt, wt, dyt, pairs, num, False, True)
        edin, edw = O.indice_conv_backward(x, w, dy, ep, en, subm=True)
        print(ks, 'din err', float(np.abs(din.cpu().numpy() - edin).max()), 'dw err', float(np.abs(dw.cpu().numpy() - edw).max()))
    except Exception as e:
        print(ks, 'FAILED', type(e).__name__, str(e)[:300])
