"""Run-to-run spread of a SIRLayer's outputs and gradients, fused (at each tile size) and per-operator path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from objectcentricocccompletion_amd import _lib as L, sir

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(7)
layer = sir.SIRLayer(in_channels=24, feat_channels=[128, 128], with_cluster_center=False, rel_mlp_hidden_dims=[16, 32],
                     rel_mlp_in_channel=13, norm_cfg=dict(type='LN', eps=1e-3), mode='max', return_point_feats=True,
                     rel_dist_scaler=10.0, xyz_normalizer=[20, 20, 4], act='gelu', dropout=0).to(dev)
M, G = 3000, 40
sizes = torch.randint(20, 130, (G,), generator=g)
inv = torch.repeat_interleave(torch.arange(G), sizes)[:M]
M = inv.numel()
feats = torch.randn(M, 24, generator=g).to(dev)
fc = torch.randn(M, 13, generator=g).to(dev)
dp, dg = torch.randn(M, 128, generator=g).to(dev), torch.randn(int(inv.max()) + 1, 256, generator=g).to(dev)
inv = inv.to(dev).int()


def run():
    layer.zero_grad(set_to_none=True)
    x = feats.clone().requires_grad_(True)
    pf, gf = layer(x, inv, fc)
    ((pf * dp).sum() + (gf * dg).sum()).backward()
    return [pf.detach(), gf.detach(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]


def spread(tag, n=30):
    first = run()
    worst = [0.0] * len(first)
    for _ in range(n):
        cur = run()
        for i, (a, b) in enumerate(zip(cur, first)):
            worst[i] = max(worst[i], float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)))
    print(tag, 'pf %.1e gf %.1e dx %.1e params max %.1e' % (worst[0], worst[1], worst[2], max(worst[3:])))
    return first


ref = {}
for tile in (16, 32, 64):
    L.check(L.lib.ococc_point_mlp_force_tile(tile), 'tile')
    ref[tile] = spread(f'fused tile {tile}')
L.check(L.lib.ococc_point_mlp_force_tile(0), 'tile')
sir.POINT_LAYER_KERNEL = False
ops = spread('operator path')
for tile in (16, 32, 64):
    d = [float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)) for a, b in zip(ref[tile], ops)]
    print(f'fused tile {tile} vs operator path: pf %.1e gf %.1e dx %.1e params max %.1e' % (d[0], d[1], d[2], max(d[3:])))
