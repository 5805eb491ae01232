"""Parity at BASELINE.json's full configs[1] size (64 grids x 40^3 cells, 2000 points each), where the CPU
oracle is too slow to be the checker: size-independent properties of the domain instead --
sortedness / idempotence of the unique, the mirror symmetry of a sub-manifold rulebook, conservation in the
scatter, linearity and the adjoint identities <conv(x), dy> = <x, dgrad(dy)> = <W, wgrad(x, dy)> of the
sparse convolution, and bit-exact agreement of the HIP-graph replay with the eager launch sequence."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, P, SHAPE = 64, 2000, [40, 40, 40]


@pytest.fixture(scope='module')
def scene(dev):
    from objectcentricocccompletion_amd.occ_encoder import synthetic_object_grids
    from objectcentricocccompletion_amd.voxel import voxelization
    xyz, feats, bidx = synthetic_object_grids(B, P, seed=5, device=dev)
    zyx = voxelization(xyz, [0.2, 0.2, 0.2], [-4, -4, -4, 4, 4, 4], -1, -1)
    coors = torch.cat([bidx.view(-1, 1).int(), zyx], 1)
    return xyz, feats, bidx, coors


def test_unique_sorted_idempotent_and_counts(dev, scene):
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    _, _, _, coors = scene
    dims = [B] + SHAPE
    uc, inv, cnt = grid_unique(coors, dims)
    key = ((uc[:, 0].long() * 40 + uc[:, 1]) * 40 + uc[:, 2]) * 40 + uc[:, 3]
    assert bool((key[1:] > key[:-1]).all())                          # strictly sorted => unique
    assert int(cnt.sum()) == coors.shape[0] and int(cnt.min()) >= 1
    assert torch.equal(uc[inv.long()], coors)                         # the inverse map reproduces the input
    uc2, inv2, cnt2 = grid_unique(uc, dims)                           # idempotence
    assert torch.equal(uc2, uc) and torch.equal(inv2, torch.arange(uc.shape[0], device=dev, dtype=torch.int32))
    assert bool((cnt2 == 1).all())
    assert torch.equal(torch.bincount(inv.long(), minlength=uc.shape[0]).int(), cnt)


def test_scatter_conservation(dev, scene):
    from objectcentricocccompletion_amd.voxel import dynamic_scatter
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    _, feats, _, coors = scene
    dims = [B] + SHAPE
    vsum, vc = dynamic_scatter(feats, coors, 'sum', grid_shape=dims)
    vmean, _ = dynamic_scatter(feats, coors, 'mean', grid_shape=dims)
    vmax, _ = dynamic_scatter(feats, coors, 'max', grid_shape=dims)
    _, inv, cnt = grid_unique(coors, dims)
    torch.testing.assert_close(vsum.double().sum(0), feats.double().sum(0), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close((vmean.double() * cnt[:, None]).sum(0), feats.double().sum(0), rtol=1e-5, atol=1e-3)
    assert bool((vmax >= vmean - 1e-5).all())                         # max dominates mean, per voxel and channel
    assert float(vmax.max()) == float(feats.max())
    single = cnt == 1                                                 # one-point voxels: all three reductions agree
    first = torch.zeros(vc.shape[0], dtype=torch.long, device=dev).scatter_(0, inv.long(), torch.arange(coors.shape[0], device=dev))
    assert torch.equal(vmax[single], feats[first[single]])
    torch.testing.assert_close(vsum[single], feats[first[single]], rtol=0, atol=0)


def test_subm_rulebook_symmetry(dev, scene):
    from objectcentricocccompletion_amd.spconv import ops
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    _, _, _, coors = scene
    uc, _, _ = grid_unique(coors, [B] + SHAPE)
    n = uc.shape[0]
    _, pairs, num = ops.get_indice_pairs(uc, B, SHAPE, 3, subm=True)
    nbr = pairs._ococc.tables[(False, 'fwd')][0]
    assert torch.equal(num, num.flip(0))                              # offset k and its mirror 26-k pair up
    assert int(num[13]) == n and torch.equal(nbr[13], torch.arange(n, device=dev, dtype=torch.int32))
    assert int(num.sum()) == int((nbr >= 0).sum())
    for k in (0, 5, 12):
        rows = torch.nonzero(nbr[k] >= 0).squeeze(1)
        src = nbr[k][rows].long()
        assert torch.equal(nbr[26 - k][src].long(), rows)             # o sees i through k  <=>  i sees o through 26-k
        off = torch.tensor([k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1], device=dev, dtype=torch.int32)
        assert torch.equal(uc[src][:, 1:], uc[rows][:, 1:] + off) and torch.equal(uc[src][:, 0], uc[rows][:, 0])
        c = int(num[k])
        assert bool((pairs[k, :, :c] >= 0).all()) and bool((pairs[k, :, c:] == -1).all())


@pytest.mark.parametrize('cin,cout', [(16, 32), (64, 128)])
def test_conv_linearity_and_adjoint_identities(dev, scene, cin, cout):
    from objectcentricocccompletion_amd.spconv import ops
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    _, _, _, coors = scene
    uc, _, _ = grid_unique(coors, [B] + SHAPE)
    n = uc.shape[0]
    _, pairs, num = ops.get_indice_pairs(uc, B, SHAPE, 3, subm=True)
    g = torch.Generator().manual_seed(cin)
    bf = lambda t: t.to(dev).bfloat16()
    x, y = bf(torch.randn(n, cin, generator=g)), bf(torch.randn(n, cin, generator=g))
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.05).to(dev).bfloat16().float()   # bf16-exact weights
    dy = bf(torch.randn(n, cout, generator=g))
    conv = lambda t: ops.indice_conv(t, w, pairs, num, n, False, True).float()
    # linearity in the input (x + y is rounded to bf16 once more: tolerance of one bf16 ulp of the sum)
    lhs, rhs = conv((x.float() + y.float()).bfloat16()), conv(x) + conv(y)
    assert float((lhs - rhs).abs().max()) <= 3e-2 * float(rhs.abs().max())
    # centre-only kernel = plain matrix product
    wc = torch.zeros_like(w)
    wc[1, 1, 1] = w[1, 1, 1]
    got = ops.indice_conv(x, wc, pairs, num, n, False, True).float()
    exp = x.float() @ w[1, 1, 1]
    assert float((got - exp).abs().max()) <= 1e-2 * float(exp.abs().max())
    # adjoint identities of the bilinear map (x, W) -> conv
    out = conv(x).double()
    din, dw = ops.indice_conv_backward(x, w, dy, pairs, num, False, True)
    a = float((out * dy.double()).sum())
    b_ = float((x.double() * din.double()).sum())
    c_ = float((w.double() * dw.double()).sum())
    scale = float(out.abs().mean() * dy.double().abs().mean() * out.numel()) ** 0.5 * float(out.numel()) ** 0.0
    assert abs(a - b_) <= 2e-3 * max(abs(a), scale) and abs(a - c_) <= 2e-3 * max(abs(a), scale), (a, b_, c_)


def test_graph_replay_matches_eager_full_size(dev, scene):
    import os
    assert os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0'
    from objectcentricocccompletion_amd.graph import GraphedStep
    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder
    xyz, feats, bidx, _ = scene
    torch.manual_seed(0)
    model = SubMOccEncoder(grouped_points=True).to(dev)
    with torch.no_grad():
        n = model(xyz, feats, bidx, B).features.shape[0]
    d = torch.zeros(xyz.shape[0], 128, dtype=torch.bfloat16, device=dev)
    d[:n] = (torch.randn(n, 128, device=dev) / n).to(torch.bfloat16)

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        out = model(xyz, feats, bidx, B, static=True)
        out.features.backward(d)
        return out

    g = GraphedStep(fwd_bwd, warmup=1)
    out_g = g.replay()
    torch.cuda.synchronize()
    feat_g = out_g.features.clone()
    grads_g = [p.grad.clone() for p in model.parameters()]
    out_e = fwd_bwd()
    assert torch.equal(feat_g, out_e.features)
    for a, p in zip(grads_g, model.parameters()):
        assert torch.equal(a, p.grad)
    assert bool((out_e.indices[n:] == -1).all()) and bool(torch.isfinite(out_e.features.float()).all())


def test_fused_front_end_properties_full_size(dev, scene):
    """ococc_voxelize_scatter_mean_f32 at the benchmark size: sorted unique rows, an inverse map that reproduces the
    cells, counts that sum to the points, conservation of the feature sums, padding rows inert."""
    from objectcentricocccompletion_amd.voxel import voxelize_scatter_mean
    xyz, feats, bidx, coors = scene
    vf, vc, inv, cnt, meta = voxelize_scatter_mean(xyz, bidx, feats, [0.2, 0.2, 0.2], [-4, -4, -4, 4, 4, 4], SHAPE, B,
                                                   static=True)
    num = int(meta[0])
    assert int(meta[1]) == 0 and vc.shape[0] == B * P and 0 < num <= B * P
    key = ((vc[:num, 0].long() * 40 + vc[:num, 1]) * 40 + vc[:num, 2]) * 40 + vc[:num, 3]
    assert bool((key[1:] > key[:-1]).all())
    assert torch.equal(vc[inv.long()], coors)
    assert int(cnt.sum()) == coors.shape[0] and int(cnt[:num].min()) >= 1
    assert bool((vc[num:] == -1).all()) and bool((cnt[num:] == 0).all()) and bool((vf[num:] == 0).all())
    torch.testing.assert_close((vf.double() * cnt[:, None]).sum(0), feats.double().sum(0), rtol=1e-5, atol=1e-3)
    vf2, vc2, _, cnt2, meta2 = voxelize_scatter_mean(xyz, bidx, feats, [0.2, 0.2, 0.2], [-4, -4, -4, 4, 4, 4], SHAPE, B)
    assert vc2.shape[0] == num and torch.equal(vc2, vc[:num]) and torch.equal(cnt2, cnt[:num])
    assert torch.equal(vf2[cnt2 <= 2], vf[:num][cnt2 <= 2])          # deterministic up to the order of >= 3 atomics


@pytest.mark.parametrize('cin,cout', [(64, 32), (128, 64)])
def test_tile_conv_adjoint_identities_full_size(dev, scene, cin, cout):
    """The compact-then-multiply kernel at the benchmark size: <conv(x), dy> = <x, dgrad(dy)> (both through the
    tile kernel: forward cin -> cout and its dgrad cout -> cin), and agreement with the output-stationary kernels."""
    from objectcentricocccompletion_amd.spconv import ops
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    _, _, _, coors = scene
    uc, _, _ = grid_unique(coors, [B] + SHAPE)
    n = uc.shape[0]
    _, pairs, num = ops.get_indice_pairs(uc, B, SHAPE, 3, subm=True)
    g = torch.Generator().manual_seed(cin + 1)
    bf = lambda t: t.to(dev).bfloat16()
    x, dy = bf(torch.randn(n, cin, generator=g)), bf(torch.randn(n, cout, generator=g))
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.05).to(dev).bfloat16().float()
    res = {}
    for tile in (False, True):
        ops.SPARSE_TILE_CONV = tile
        try:
            y = ops.indice_conv(x, w, pairs, num, n, False, True).float()
            dx, _ = ops.indice_conv_backward(x, w, dy, pairs, num, False, True)
        finally:
            ops.SPARSE_TILE_CONV = None
        res[tile] = (y, dx.float())
    for a, b in zip(res[False], res[True]):
        assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max())
    y, dx = res[True]
    lhs = float((y.double() * dy.double()).sum())
    rhs = float((x.double() * dx.double()).sum())
    assert abs(lhs - rhs) <= 2e-2 * max(abs(lhs), abs(rhs), 1.0) + 1e-2 * float(y.abs().max()) * n ** 0.5


def test_tile_conv_layernorm_epilogue_full_size(dev, scene):
    """The 32 -> 64 layer of the benchmark with LayerNorm + GELU in the tile kernel's epilogue, at full size:
    run-to-run identical, conv output identical to the plain tile kernel, every output row normalised (the
    pre-activation recomputed from the returned statistics has zero mean / unit variance per row), and the
    statistics equal to those of the separate LayerNorm kernel."""
    from objectcentricocccompletion_amd import _lib as L
    from objectcentricocccompletion_amd.spconv import ops
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique
    _, _, _, coors = scene
    uc, _, _ = grid_unique(coors, [B] + SHAPE)
    n = uc.shape[0]
    _, pairs, num = ops.get_indice_pairs(uc, B, SHAPE, 3, subm=True)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, 32, generator=g).to(dev).bfloat16()
    w = (torch.randn(3, 3, 3, 32, 64, generator=g) * 0.05).to(dev)
    gamma = torch.ones(64, device=dev)
    beta = torch.zeros(64, device=dev)
    ops.SPARSE_TILE_CONV = True
    try:
        conv = ops.indice_conv(x, w, pairs, num, n, False, True)
        a = ops.indice_conv_ln(x, w, gamma, beta, 1e-3, 0, pairs, num, n, False, True)
        b = ops.indice_conv_ln(x, w, gamma, beta, 1e-3, 0, pairs, num, n, False, True)
    finally:
        ops.SPARSE_TILE_CONV = None
    for t, u in zip(a, b):
        assert torch.equal(t, u)
    conv_out, y, stats = a
    assert torch.equal(conv_out, conv)
    z = (conv_out.float() - stats[:, :1]) * stats[:, 1:]              # act = none, gamma = 1, beta = 0: z == y
    assert float(z.mean(1).abs().max()) <= 1e-4
    var = z.var(1, unbiased=False)
    nz = conv_out.float().var(1, unbiased=False) > 1e-2              # (rows of ~zero variance are dominated by eps)
    assert float((var[nz] - 1).abs().max()) <= 0.1
    assert float((y.float() - z).abs().max()) <= 2e-2 * float(z.abs().max())
    ref_y = torch.empty_like(conv)
    ref_stats = torch.empty_like(stats)
    L.check(L.lib.ococc_layernorm_act_fwd(L.ptr(conv), n, 64, L.ptr(gamma), L.ptr(beta), 1e-3, 0, L.ptr(ref_y),
                                          L.ptr(ref_stats), L.BF16, L.stream()), 'ln')
    torch.cuda.synchronize()
    assert float((stats - ref_stats).abs().max()) <= 1e-4 * float(ref_stats.abs().max())
