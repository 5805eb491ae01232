"""GPU parity of the fused per-point layer (csrc/point_mlp.hip, A6) against float64 torch math of the reference's chain
Linear(bias=False) -> LayerNorm -> GELU -> scatter max with the gather-back / concatenation / product in front
(voxel_encoder.py:764-832), forward and backward, at the channel widths the ococcnet config uses (odd ones included),
and of the whole SIRLayer: fused realisation against the per-operator one."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[16, 32, 64], ids=['tile16', 'tile32', 'tile64'])
def tile_rows(request):
    """every test of this file runs with the fwd / bwd launches pinned to 32-row and to 64-row tiles (the library picks
    by input size otherwise: small inputs, as these tests' are, would never reach the 64-row instantiations)"""
    from objectcentricocccompletion_amd import _lib as L
    L.check(L.lib.ococc_point_mlp_force_tile(request.param), 'force_tile')
    yield request.param
    L.check(L.lib.ococc_point_mlp_force_tile(0), 'force_tile')


def _ref_layer(a, w, g, be, eps, act, mul, cs, b, bs, v, inv, G, seg_max):
    x = a
    if mul is not None:
        x = x * mul
    if cs is not None:
        x = x * cs
    parts = [x]
    if b is not None:
        parts.append(b * bs)
    if v is not None:
        parts.append(v[inv.long()])
    x = torch.cat(parts, 1)
    z = x @ w.t()
    if g is not None:
        z = torch.nn.functional.layer_norm(z, (w.shape[0],), g, be, eps)
    y = torch.nn.functional.gelu(z) if act == 'gelu' else (z.clamp(min=0) if act == 'relu' else z)
    if not seg_max:
        return y, None
    m = torch.full((G, w.shape[0]), float('-inf'), dtype=y.dtype, device=y.device)
    m = m.scatter_reduce(0, inv.long()[:, None].expand_as(y), y, 'amax')
    return y, m


@pytest.mark.parametrize('shape', [
    # (rows, ka, mul?, colscale?, kb, kv, n, ln, act, seg_max)
    (1000, 13, False, True, 0, 0, 16, True, 'gelu', False),     # rel_mlp layer 0 (RoI encoder)
    (1000, 16, False, False, 0, 0, 32, True, 'gelu', False),
    (777, 32, False, False, 0, 0, 24, True, 'gelu', False),     # odd output width
    (1000, 32, False, False, 0, 0, 144, True, 'gelu', False),
    (700, 3, False, True, 0, 0, 16, True, 'gelu', False),
    (900, 32, False, False, 0, 0, 131, True, 'gelu', False),    # 131 outputs: padding inside the last block
    (1000, 24, True, True, 0, 0, 128, True, 'gelu', True),      # vfe 0, RoI encoder block 0
    (1000, 144, True, True, 0, 0, 128, True, 'gelu', True),
    (1000, 15, True, True, 3, 0, 128, True, 'gelu', True),      # vfe 0, AE encoder (cluster centre appended)
    (1000, 131, True, True, 3, 0, 128, True, 'gelu', True),
    (1000, 128, False, False, 0, 128, 128, True, 'gelu', True), # vfe 1: gather-back
    (130, 128, False, False, 0, 128, 128, True, 'relu', True),
    (64, 20, False, False, 0, 0, 48, False, 'none', False),     # no norm
    (5, 8, True, False, 2, 4, 16, True, 'gelu', True),          # fewer rows than a tile
])
def test_point_layer_vs_torch_f64(dev, shape):
    from objectcentricocccompletion_amd.point_mlp import point_layer
    rows, ka, has_mul, has_cs, kb, kv, n, ln, act, seg_max = shape
    g = torch.Generator().manual_seed(sum(shape[:2]) + n)
    G = max(1, rows // 37)
    sizes = torch.randint(1, 74, (G,), generator=g)
    inv = torch.repeat_interleave(torch.arange(G), sizes)[:rows]
    if inv.numel() < rows:
        inv = torch.cat([inv, torch.full((rows - inv.numel(),), G - 1)])
    G = int(inv.max()) + 1
    R = lambda *s: torch.randn(*s, generator=g)
    a, mul = R(rows, ka), (R(rows, ka) if has_mul else None)
    cs = (torch.rand(ka, generator=g) + 0.5) if has_cs else None
    b, v = (R(rows, kb) if kb else None), (R(G, kv) if kv else None)
    w = R(n, ka + kb + kv) / (ka + kb + kv) ** 0.5
    gam, bet = ((1 + 0.2 * R(n)), 0.1 * R(n)) if ln else (None, None)
    dy, dm = R(rows, n), R(G, n)
    bs = 0.37

    def leaf(t, dt):
        return None if t is None else t.to(dev, dt).requires_grad_(True)
    outs = {}
    for name, dt in (('hip', torch.float32), ('ref', torch.float64)):
        A, M, B, V, W, Gm, Bt = (leaf(t, dt) for t in (a, mul, b, v, w, gam, bet))
        csd = None if cs is None else cs.to(dev, dt)
        invd = inv.to(dev, torch.int32)
        if name == 'hip':
            res = point_layer(A, W, Gm, Bt, 1e-3, act, mul=M, colscale=csd, b=B, bscale=bs, v=V, inv=invd if (kv or seg_max) else None,
                              num_segments=G, seg_max=seg_max)
            y, m = res if seg_max else (res, None)
        else:
            y, m = _ref_layer(A, W, Gm, Bt, 1e-3, act, M, csd, B, bs, V, invd, G, seg_max)
        loss = (y * dy.to(dev, dt)).sum() + ((m * dm.to(dev, dt)).sum() if seg_max else 0)
        loss.backward()
        outs[name] = dict(y=y.detach(), m=None if m is None else m.detach(),
                          grads=[None if t is None else t.grad for t in (A, M, B, V, W, Gm, Bt)])
    rel = lambda x, r: float((x.double() - r).abs().max() / r.abs().max().clamp(min=1e-30))
    assert rel(outs['hip']['y'], outs['ref']['y']) < 2e-5
    if seg_max:
        assert rel(outs['hip']['m'], outs['ref']['m']) < 2e-5
    for gh, gr, nm in zip(outs['hip']['grads'], outs['ref']['grads'], ('a', 'mul', 'b', 'v', 'w', 'gamma', 'beta')):
        if gr is None:
            continue
        assert gh is not None, nm
        assert rel(gh, gr) < 2e-4, (nm, rel(gh, gr))


def test_segment_max_ties_go_to_the_smallest_row(dev):
    """Equal maxima inside a segment (duplicated points): the gradient of the maximum goes to the smallest row, the rule
    of the reference's DynamicScatter (scatter_points_cuda.cu:136-160) that segment_reduce follows too."""
    from objectcentricocccompletion_amd.point_mlp import point_layer
    g = torch.Generator().manual_seed(3)
    rows, k, n = 200, 8, 16
    a = torch.randn(rows, k, generator=g)
    a[50:60] = a[50]                    # ten identical rows inside segment 1, spread over one tile
    a[130:140] = a[70]                  # and a copy of row 70 in the same segment but another tile
    inv = torch.cat([torch.zeros(40), torch.ones(110), torch.full((50,), 2)]).int()
    w = torch.randn(n, k, generator=g)
    A = a.to(dev).requires_grad_(True)
    y, m = point_layer(A, w.to(dev), None, None, 0.0, 'none', inv=inv.to(dev), num_segments=3, seg_max=True)
    m.sum().backward()
    yc = y.detach().cpu()
    got = (A.grad.abs().sum(1) > 0).cpu()
    for seg in range(3):
        rows_of = torch.nonzero(inv == seg).flatten()
        for ch in range(n):
            col = yc[rows_of, ch]
            first = int(rows_of[int(torch.nonzero(col == col.max()).flatten()[0])])
            assert bool(got[first])
    assert not bool(got[51:60].any()) and not bool(got[130:140].any())


@pytest.mark.parametrize('cfg', [dict(in_channels=24, rel_in=13, cluster=False), dict(in_channels=15, rel_in=3, cluster=True)])
def test_sir_layer_fused_equals_operator_path(dev, cfg):
    from objectcentricocccompletion_amd import sir
    g = torch.Generator().manual_seed(7)
    layer = sir.SIRLayer(in_channels=cfg['in_channels'], feat_channels=[128, 128], with_cluster_center=cfg['cluster'],
                         rel_mlp_hidden_dims=[16, 32], rel_mlp_in_channel=cfg['rel_in'], norm_cfg=dict(type='LN', eps=1e-3),
                         mode='max', return_point_feats=True, rel_dist_scaler=10.0, xyz_normalizer=[20, 20, 4], act='gelu',
                         dropout=0).to(dev)
    assert layer._fusable()
    M, G = 3000, 40
    sizes = torch.randint(20, 130, (G,), generator=g)
    inv = torch.repeat_interleave(torch.arange(G), sizes)[:M]
    M = inv.numel()
    feats = torch.randn(M, cfg['in_channels'], generator=g).to(dev)
    fc = torch.randn(M, 13, generator=g).to(dev) if not cfg['cluster'] else None
    dp, dg = torch.randn(M, 128, generator=g).to(dev), torch.randn(int(inv.max()) + 1, 256, generator=g).to(dev)
    runs = []
    for fused in (True, False):
        sir.POINT_LAYER_KERNEL = fused
        try:
            layer.zero_grad(set_to_none=True)
            x = feats.clone().requires_grad_(True)
            pf, gf = layer(x, inv.to(dev).int(), fc)
            ((pf * dp).sum() + (gf * dg).sum()).backward()
            runs.append((pf.detach(), gf.detach(), x.grad.clone(), [p.grad.clone() for p in layer.parameters()]))
        finally:
            sir.POINT_LAYER_KERNEL = True
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
    a, b = runs
    # (input gradient: norm-wise.  Where two rows of a group agree in a channel to the last bits, the two paths -- whose y
    # differ by f32 summation order -- may name different rows the maximum and route that gradient to different rows: an
    # element-wise bound then fails once in a dozen runs by a few 1e-3 of the largest entry, on both sides legitimately)
    nrel = lambda a, b: float((a - b).norm() / b.norm().clamp(min=1e-30))
    assert rel(a[0], b[0]) < 1e-4 and rel(a[1], b[1]) < 1e-4 and nrel(a[2], b[2]) < 1e-3
    for p, q in zip(a[3], b[3]):
        assert rel(p, q) < 1e-3


@pytest.mark.parametrize('use', ['both', 'groups', 'points'])
@pytest.mark.parametrize('in_channels', [131, 24])
def test_sir_layer_as_one_autograd_node(dev, in_channels, use):
    """The whole-layer node (one Function per SIRLayer, the shortcut inside) against the chain of per-block Functions: the
    same launches in the same order, so outputs and gradients agree to the bit; and against the per-operator path."""
    from objectcentricocccompletion_amd import sir
    g = torch.Generator().manual_seed(11)
    layer = sir.SIRLayer(in_channels=in_channels, feat_channels=[128, 128], with_cluster_center=False,
                         rel_mlp_hidden_dims=[16, 32], rel_mlp_in_channel=13, norm_cfg=dict(type='LN', eps=1e-3),
                         mode='max', return_point_feats=True, rel_dist_scaler=10.0, xyz_normalizer=[20, 20, 4], act='gelu',
                         dropout=0).to(dev)
    G = 37
    sizes = torch.randint(1, 150, (G,), generator=g)
    inv = torch.repeat_interleave(torch.arange(G), sizes).to(dev)
    M = inv.numel()
    feats = torch.randn(M, in_channels, generator=g).to(dev)
    fc = torch.randn(M, 13, generator=g).to(dev)
    dp, dg = torch.randn(M, 128, generator=g).to(dev), torch.randn(G, 256, generator=g).to(dev)

    def run(whole, kernel=True, native=False):
        sir.WHOLE_LAYER_NODE, sir.POINT_LAYER_KERNEL, sir.NATIVE_LAYER = whole, kernel, native
        try:
            layer.zero_grad(set_to_none=True)
            x = feats.clone().requires_grad_(True)
            pf, gf = layer(x, inv, fc)
            loss = 0
            if use in ('both', 'points'):
                loss = loss + (pf * dp).sum()
            if use in ('both', 'groups'):
                loss = loss + (gf * dg).sum()
            loss.backward()
            return pf.detach(), gf.detach(), x.grad.clone(), [None if p.grad is None else p.grad.clone() for p in layer.parameters()]
        finally:
            sir.WHOLE_LAYER_NODE, sir.POINT_LAYER_KERNEL, sir.NATIVE_LAYER = True, True, True

    a, b, c = run(True), run(False), run(True, kernel=False)
    nat = run(True, native=True)   # the same launches issued by the library (csrc/sir_layer.hip)
    assert torch.equal(nat[0], a[0]) and torch.equal(nat[1], a[1])
    assert float((nat[2] - a[2]).abs().max()) <= 1e-6 * float(a[2].abs().max())
    for p, q in zip(nat[3], a[3]):
        assert (p is None) == (q is None)
        if p is not None:
            assert float((p - q).abs().max()) <= 1e-5 * float(q.abs().max().clamp(min=1e-30))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    rel = lambda p, q: float((p - q).abs().max() / q.abs().max().clamp(min=1e-30))
    # (the chain adds the shortcut's and the gathered maxima's gradients through autograd's accumulation: another order)
    assert rel(a[2], b[2]) < 1e-6
    for p, q in zip(a[3], b[3]):
        assert (p is None) == (q is None)
        if p is not None:
            assert rel(p, q) < 1e-5
    nrel = lambda p, q: float((p - q).norm() / q.norm().clamp(min=1e-30))   # (see test_sir_layer_fused_equals_operator_path)
    assert rel(a[0], c[0]) < 1e-4 and rel(a[1], c[1]) < 1e-4 and nrel(a[2], c[2]) < 1e-3
    for p, q in zip(a[3], c[3]):
        if p is not None and q is not None:
            assert rel(p, q) < 1e-3


@pytest.mark.parametrize('rows', [40, 3000, 40000])
@pytest.mark.parametrize('cfg', [dict(in_channels=24, rel_in=13, cluster=False), dict(in_channels=144, rel_in=13, cluster=False),
                                 dict(in_channels=15, rel_in=3, cluster=True), dict(in_channels=131, rel_in=3, cluster=True)])
@pytest.mark.parametrize('use', ['both', 'groups', 'sliced'])
def test_sir_layer_in_one_launch_equals_the_launch_per_block(dev, cfg, rows, use):
    """ococc_sir_layer_{fwd,bwd}_f32 as ONE launch per direction (csrc/sir_fused_impl.hpp: persistent grid, grid-wide
    barriers where the segment maxima cross tiles) against the same calls issuing one launch per block: the tile bodies
    are the same code, so the forward results agree to the bit; in the backward pass only the float atomics that sum a
    group's gradient over its tiles arrive in another order.  40 000 rows at 16-row tiles = 2 500 tiles on at most 512
    resident workgroups: several tiles per workgroup.  Groups of 1 .. 300 rows span up to 20 tiles; duplicated rows tie
    for the maximum (the smallest row must win in both forms).  The four SIRLayer shapes of configs[2]."""
    from objectcentricocccompletion_amd import _lib as L, sir
    g = torch.Generator().manual_seed(13)
    layer = sir.SIRLayer(in_channels=cfg['in_channels'], feat_channels=[128, 128], with_cluster_center=cfg['cluster'],
                         rel_mlp_hidden_dims=[16, 32], rel_mlp_in_channel=cfg['rel_in'], norm_cfg=dict(type='LN', eps=1e-3),
                         mode='max', return_point_feats=True, rel_dist_scaler=10.0, xyz_normalizer=[20, 20, 4], act='gelu',
                         dropout=0).to(dev)
    sizes = torch.randint(1, 300, (rows // 100 + 2,), generator=g)
    inv = torch.repeat_interleave(torch.arange(sizes.numel()), sizes)[:rows].to(dev)
    M, G = inv.numel(), int(inv.max()) + 1
    feats = torch.randn(M, cfg['in_channels'], generator=g)
    if M > 30:
        feats[10:14] = feats[10]          # ties inside a group ...
        feats[M - 3:] = feats[M - 3]      # ... and in the last tile
    feats = feats.to(dev)
    # (the cluster offsets are handed in: derived inside the layer they come from a segment mean whose float atomics differ
    # from run to run in the last bit)
    fc = torch.randn(M, cfg['rel_in'], generator=g).to(dev)
    dp, dg = torch.randn(M, 128, generator=g).to(dev), torch.randn(G, 256, generator=g).to(dev)

    def run(fused):
        L.check(L.lib.ococc_sir_layer_set_fused(int(fused)), 'set_fused')
        try:
            layer.zero_grad(set_to_none=True)
            x = feats.clone().requires_grad_(True)
            pf, gf = layer(x, inv, fc)
            if use == 'sliced':   # the gradients arrive as column slices of wider tensors (the concatenations around a
                # layer in SIR.forward): read in place through their row strides, no copy
                loss = (torch.cat([gf, gf.detach()], 1) * torch.cat([dg, dg], 1)).sum() + \
                       (torch.cat([pf.detach()[:, :3], pf], 1) * torch.cat([dp[:, :3], dp], 1)).sum()
            else:
                loss = (gf * dg).sum()
                if use == 'both':
                    loss = loss + (pf * dp).sum()
            loss.backward()
            import ctypes
            status = ctypes.c_int32(-1)
            L.check(L.lib.ococc_sir_layer_fused_status(L.stream(), ctypes.byref(status)), 'fused_status')
            assert status.value == 0, f'a grid barrier of the one-launch layer did not complete (barrier {status.value - 1})'
            return pf.detach(), gf.detach(), x.grad.clone(), [p.grad.clone() for p in layer.parameters()]
        finally:
            L.check(L.lib.ococc_sir_layer_set_fused(-1), 'set_fused')

    one = run(True)
    per_block = run(False)
    assert torch.equal(one[0], per_block[0]) and torch.equal(one[1], per_block[1])
    rel = lambda p, q: float((p - q).abs().max() / q.abs().max().clamp(min=1e-30))
    assert rel(one[2], per_block[2]) < 1e-5, rel(one[2], per_block[2])
    for p, q in zip(one[3], per_block[3]):
        assert rel(p, q) < 1e-5, rel(p, q)


def test_a_stranded_grid_barrier_raises_in_the_product_path(dev):
    """A one-launch SIRLayer whose persistent grid is not resident as a whole (here: forced to 8192 workgroups, several
    times what the device holds; in the field: two processes or streams with such grids on one device) cannot complete its
    grid barriers.  The bounded wait gives up, the kernel says so in a host-mapped word, and the PRODUCT path raises at the
    next layer (sir.SIRLayer.forward -> ococc_sir_layer_fwd_f32 -> OCOCC_ESTRANDED) or at the caller's next check
    (sir.check_barriers, as point_pool / tools/train.py / bench.py call it) -- never a finite loss on incomplete maxima.
    After the report the process runs the per-block launches, whose results equal the one-launch form's to the bit."""
    import ctypes
    from objectcentricocccompletion_amd import _lib as L, sir
    g = torch.Generator().manual_seed(5)
    layer = sir.SIRLayer(in_channels=24, feat_channels=[128, 128], with_cluster_center=False, rel_mlp_hidden_dims=[16, 32],
                         rel_mlp_in_channel=13, norm_cfg=dict(type='LN', eps=1e-3), mode='max', return_point_feats=True,
                         rel_dist_scaler=10.0, xyz_normalizer=[20, 20, 4], act='gelu', dropout=0).to(dev)
    rows = 3000
    inv = torch.sort(torch.randint(0, 40, (rows,), generator=g)).values.to(dev)
    feats, fc = torch.randn(rows, 24, generator=g).to(dev), torch.randn(rows, 13, generator=g).to(dev)
    with torch.no_grad():
        good = layer(feats, inv, fc)   # (also sizes the persistent grid: the residency census runs once per kernel)
    torch.cuda.synchronize()
    sir.check_barriers()
    try:
        L.check(L.lib.ococc_sir_layer_set_fused(1), 'set_fused')
        L.check(L.lib.ococc_sir_layer_fused_debug(8192, 20), 'fused_debug')
        with torch.no_grad():
            layer(feats, inv, fc)      # strands: 8192 workgroups are never resident together
        torch.cuda.synchronize()
        status = ctypes.c_int32(-1)
        L.check(L.lib.ococc_sir_layer_fused_status(L.stream(), ctypes.byref(status)), 'fused_status')
        assert status.value != 0
        L.check(L.lib.ococc_sir_layer_fused_debug(0, 20), 'fused_debug')   # (the next launch would fit -- it must not happen)
        with pytest.raises(L.OcoccError, match='gave up'):
            with torch.no_grad():
                layer(feats, inv, fc)
        sir.check_barriers()           # reported once
        with torch.no_grad():
            after = layer(feats, inv, fc)   # the per-block launches from here on
        torch.cuda.synchronize()
        assert torch.equal(after[0], good[0]) and torch.equal(after[1], good[1])
        # ... and the caller-side check alone (the last layer of a pass has no next layer to report it)
        L.check(L.lib.ococc_sir_layer_set_fused(1), 'set_fused')
        L.check(L.lib.ococc_sir_layer_fused_debug(8192, 20), 'fused_debug')
        with torch.no_grad():
            layer(feats, inv, fc)
        torch.cuda.synchronize()
        with pytest.raises(L.OcoccError, match='gave up'):
            sir.check_barriers()
    finally:
        L.check(L.lib.ococc_sir_layer_fused_debug(0, 0), 'fused_debug')
        L.check(L.lib.ococc_sir_layer_set_fused(-1), 'set_fused')
    with torch.no_grad():
        again = layer(feats, inv, fc)       # the one-launch form again, whole
    torch.cuda.synchronize()
    sir.check_barriers()
    status = ctypes.c_int32(-1)
    L.check(L.lib.ococc_sir_layer_fused_status(L.stream(), ctypes.byref(status)), 'fused_status')
    assert status.value == 0 and torch.equal(again[0], good[0]) and torch.equal(again[1], good[1])


@pytest.mark.parametrize('rows', [60, 5000])
@pytest.mark.parametrize('stack', ['sir', 'head'])
def test_rel_mlp_chains_of_a_stack_in_one_launch(dev, stack, rows):
    """The rel_mlp gates of all blocks of a SIR stack from the cluster offsets they share -- one launch per direction
    (csrc/sir_rel_chains.hip, sir.rel_gates) -- against every block running its own rel_mlp: the same tile bodies, so the
    forward results agree to the bit and the gradients up to the order of the float atomics.  The two stack shapes of
    configs[2]: the SIR backbone of the auto-encoder (15 | 131 input columns, 3 offset columns) and the blocks of
    OccBBoxHead.roi_encode (24 | 144 input columns with the geometry columns, 13 offset columns)."""
    from objectcentricocccompletion_amd import sir
    g = torch.Generator().manual_seed(21)
    nb = 3
    if stack == 'sir':
        net = sir.SIR(num_blocks=nb, in_channels=[15] + [131] * (nb - 1), feat_channels=[[128, 128]] * nb,
                      rel_mlp_hidden_dims=[[16, 32]] * nb, with_rel_mlp=True, with_cluster_center=False, with_distance=False,
                      norm_cfg=dict(type='LN', eps=1e-3), mode='max', xyz_normalizer=[1, 1, 1], act='gelu', dropout=0,
                      unique_once=True).to(dev)
    else:
        net = torch.nn.ModuleList([sir.SIRLayer(in_channels=c, feat_channels=[128, 128], with_cluster_center=False,
                                                rel_mlp_hidden_dims=[16, 32], rel_mlp_in_channel=13, norm_cfg=dict(type='LN', eps=1e-3),
                                                mode='max', return_point_feats=i != nb - 1, rel_dist_scaler=10.0,
                                                xyz_normalizer=[20, 20, 4], act='gelu', dropout=0)
                                   for i, c in enumerate([24] + [144] * (nb - 1))]).to(dev)
    sizes = torch.randint(1, 90, (rows // 30 + 2,), generator=g)
    inv = torch.repeat_interleave(torch.arange(sizes.numel()), sizes)[:rows]
    M = inv.numel()
    coors = inv.to(dev).int()
    pts = torch.randn(M, 3, generator=g).to(dev)
    feats = torch.randn(M, 12 if stack == 'sir' else 8, generator=g).to(dev)
    fc = torch.randn(M, 13, generator=g).to(dev)

    def run(batched):
        sir.BATCH_REL_CHAINS = batched
        try:
            net.zero_grad(set_to_none=True)
            x = feats.clone().requires_grad_(True)
            if stack == 'sir':
                # (the offsets handed in: derived inside they come from a segment mean whose float atomics differ run to run)
                pf, gf, _ = net(pts, x, coors[:, None], f_cluster=fc[:, :3].contiguous(), dims=[int(inv.max()) + 1])
                outs = [pf, gf]
            else:
                from objectcentricocccompletion_amd.sst.sst_ops import unique_with_inverse
                new_coors, unq = unique_with_inverse(coors, [int(inv.max()) + 1])
                gates = sir.rel_gates(net, fc)
                assert (gates is not None) == batched
                cur, outs = x, []
                for i, block in enumerate(net):
                    fin = torch.cat([pts, cur, fc / 10], 1)
                    out = block(fin, coors, fc, unq_inv_once=unq, new_coors_once=new_coors, **({} if gates is None else {'gate': gates[i]}))
                    if i < nb - 1:
                        cur = out[0]
                    outs.append(out[1] if i < nb - 1 else out[0])
            loss = sum((o * torch.linspace(0.5, 1.5, o.shape[1], device=dev)).sum() for o in outs)
            loss.backward()
            return [o.detach() for o in outs], x.grad.clone(), [p.grad.clone() for p in net.parameters()]
        finally:
            sir.BATCH_REL_CHAINS = True

    a, b = run(True), run(False)
    for p, q in zip(a[0], b[0]):
        assert torch.equal(p, q)
    rel = lambda p, q: float((p - q).abs().max() / q.abs().max().clamp(min=1e-30))
    assert rel(a[1], b[1]) < 1e-5
    for p, q in zip(a[2], b[2]):
        assert rel(p, q) < 1e-5


def test_rel_gates_in_front_of_the_operator_path(dev):
    """Above POINT_LAYER_MAX_ROWS the vfe blocks of a layer run operator by operator (library GEMMs); the rel_mlp gates of the
    stack still come from the one launch (three blocks of <= 32 input channels are skinny products for the library).  Same
    function as the stack with every rel_mlp run operator by operator, to the accuracy of two f32 realisations."""
    from objectcentricocccompletion_amd import sir
    g = torch.Generator().manual_seed(5)
    nb = 3
    net = sir.SIR(num_blocks=nb, in_channels=[15] + [131] * (nb - 1), feat_channels=[[128, 128]] * nb,
                  rel_mlp_hidden_dims=[[16, 32]] * nb, with_rel_mlp=True, with_cluster_center=False, with_distance=False,
                  norm_cfg=dict(type='LN', eps=1e-3), mode='max', xyz_normalizer=[1, 1, 1], act='gelu', dropout=0,
                  unique_once=True).to(dev)
    sizes = torch.randint(1, 90, (60,), generator=g)
    inv = torch.repeat_interleave(torch.arange(sizes.numel()), sizes)
    M = inv.numel()
    coors = inv.to(dev).int()
    pts, feats = torch.randn(M, 3, generator=g).to(dev), torch.randn(M, 12, generator=g).to(dev)
    fc = torch.randn(M, 3, generator=g).to(dev)
    keep = sir.POINT_LAYER_MAX_ROWS

    def run(batched):
        sir.BATCH_REL_CHAINS, sir.POINT_LAYER_MAX_ROWS = batched, 10
        try:
            net.zero_grad(set_to_none=True)
            x = feats.clone().requires_grad_(True)
            pf, gf, _ = net(pts, x, coors[:, None], f_cluster=fc, dims=[int(inv.max()) + 1])
            ((pf * 0.7).sum() + gf.sum()).backward()
            return pf.detach(), gf.detach(), x.grad.clone(), [p.grad.clone() for p in net.parameters()]
        finally:
            sir.BATCH_REL_CHAINS, sir.POINT_LAYER_MAX_ROWS = True, keep

    a, b = run(True), run(False)
    rel = lambda p, q: float((p - q).norm() / q.norm().clamp(min=1e-30))
    assert rel(a[0], b[0]) < 1e-4 and rel(a[1], b[1]) < 1e-4 and rel(a[2], b[2]) < 1e-3
    for p, q in zip(a[3], b[3]):
        assert rel(p, q) < 1e-3


@pytest.mark.parametrize('rows,n,k', [(1, 16, 13), (31, 32, 3), (300, 128, 131), (5000, 144, 256), (70001, 64, 24),
                                       (8192, 128, 259 - 3)])
def test_weight_gradient_kernel(dev, rows, n, k):
    """ococc_point_mlp_wgrad_f32: the per-slice products sum to dz^T x_cat (f32 MFMA against an f64 product)."""
    from objectcentricocccompletion_amd import _lib as L
    g = torch.Generator().manual_seed(rows + n + k)
    dz, xc = torch.randn(rows, n, generator=g).to(dev), torch.randn(rows, k, generator=g).to(dev)
    slices = int(L.lib.ococc_point_mlp_wgrad_slices(rows))
    assert 1 <= slices <= 64
    partial = torch.full((slices, n, k), float('nan'), device=dev)
    L.check(L.lib.ococc_point_mlp_wgrad_f32(dz.data_ptr(), xc.data_ptr(), rows, n, k, partial.data_ptr(), L.stream()), 'wgrad')
    got = partial.double().sum(0)
    exp = dz.double().t() @ xc.double()
    assert bool(torch.isfinite(got).all())
    assert float((got - exp).abs().max()) <= 2e-6 * float(exp.abs().max()) * max(1.0, rows ** 0.5 / 8)


@pytest.mark.parametrize('rows', [1, 300, 8192, 70001])
def test_weight_gradients_of_several_layers_in_one_launch(dev, rows):
    """ococc_point_mlp_wgrad_multi_f32 (the blocks of one SIR layer's backward): every layer's slices bit for bit those of
    its own ococc_point_mlp_wgrad_f32 launch."""
    import ctypes
    from objectcentricocccompletion_amd import _lib as L
    shapes = [(16, 13), (32, 32), (128, 131), (144, 256), (64, 24), (128, 64), (8, 3), (70, 70)]
    g = torch.Generator().manual_seed(rows)
    slices = int(L.lib.ococc_point_mlp_wgrad_slices(rows))
    for count in (1, 5, 8):
        dz = [torch.randn(rows, n, generator=g).to(dev) for n, _ in shapes[:count]]
        xc = [torch.randn(rows, k, generator=g).to(dev) for _, k in shapes[:count]]
        one = [torch.full((slices, n, k), float('nan'), device=dev) for n, k in shapes[:count]]
        many = [torch.full((slices, n, k), float('nan'), device=dev) for n, k in shapes[:count]]
        for j, (n, k) in enumerate(shapes[:count]):
            L.check(L.lib.ococc_point_mlp_wgrad_f32(dz[j].data_ptr(), xc[j].data_ptr(), rows, n, k, one[j].data_ptr(), L.stream()), 'wgrad')
        ptrs = lambda ts: (ctypes.c_void_p * count)(*[t.data_ptr() for t in ts])
        ints = lambda vs: (ctypes.c_int32 * count)(*vs)
        L.check(L.lib.ococc_point_mlp_wgrad_multi_f32(count, ptrs(dz), ptrs(xc), rows, ints([n for n, _ in shapes[:count]]),
                                                       ints([k for _, k in shapes[:count]]), ptrs(many), L.stream()), 'wgrad_multi')
        torch.cuda.synchronize()
        for a_, b_ in zip(one, many):
            assert torch.equal(a_, b_)
    assert L.lib.ococc_point_mlp_wgrad_multi_f32(9, None, None, rows, None, None, None, None) != 0
