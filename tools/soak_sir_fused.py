"""Soak of the one-launch SIRLayer (csrc/sir_fused_impl.hpp): thousands of forward + backward launches at the
configs[2] size (one tile per workgroup) and with several tiles per workgroup, the forward compared with the first
iteration's (the kernels are deterministic in the forward), the barrier status word checked at the end.
GPU box:  python tools/soak_sir_fused.py [iterations=3000]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from objectcentricocccompletion_amd import _lib as L, sir  # noqa: E402


def status():
    s = ctypes.c_int32(-1)
    L.check(L.lib.ococc_sir_layer_fused_status(L.stream(), ctypes.byref(s)), 'fused_status')
    return s.value


def run(rows, iters, force):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    layer = sir.SIRLayer(in_channels=144, feat_channels=[128, 128], with_cluster_center=False, rel_mlp_hidden_dims=[16, 32],
                         rel_mlp_in_channel=13, norm_cfg=dict(type='LN', eps=1e-3), mode='max', return_point_feats=True,
                         rel_dist_scaler=10.0, xyz_normalizer=[20, 20, 4], act='gelu', dropout=0).to(dev)
    sizes = torch.randint(1, 130, (rows // 40 + 2,), generator=g)
    inv = torch.repeat_interleave(torch.arange(sizes.numel()), sizes)[:rows].to(dev)
    M, G = inv.numel(), int(inv.max()) + 1
    feats, fc = torch.randn(M, 144, generator=g).to(dev), torch.randn(M, 13, generator=g).to(dev)
    dp, dg = torch.randn(M, 128, generator=g).to(dev), torch.randn(G, 256, generator=g).to(dev)
    L.check(L.lib.ococc_sir_layer_set_fused(1 if force else -1), 'set_fused')
    first = None
    t0 = time.time()
    for it in range(iters):
        layer.zero_grad(set_to_none=True)
        x = feats.clone().requires_grad_(True)
        pf, gf = layer(x, inv, fc)
        ((pf * dp).sum() + (gf * dg).sum()).backward()
        if it % 200 == 0:
            if first is None:
                first = (pf.detach().clone(), gf.detach().clone())
            assert torch.equal(pf, first[0]) and torch.equal(gf, first[1]), f'forward changed at iteration {it}'
            assert bool(torch.isfinite(x.grad).all())
    torch.cuda.synchronize()
    st = status()
    L.check(L.lib.ococc_sir_layer_set_fused(-1), 'set_fused')
    print(f'{rows} rows, {iters} iterations ({"several tiles per workgroup" if force else "default policy"}): '
          f'{(time.time() - t0) / iters * 1e3:.3f} ms per forward + backward, barrier status {st}')
    assert st == 0


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    run(8192, n, False)
    run(2000, n, False)
    run(60000, max(n // 6, 10), True)
