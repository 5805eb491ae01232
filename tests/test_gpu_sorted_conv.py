"""ococc_sparse_conv_sorted_bf16 / ococc_subm_row_order (sub-manifold convolution with the output rows taken in
neighbour-pattern order) against the voxel-order output-stationary kernel, which is pinned to the oracle in
test_gpu_spconv.py: the two must agree BIT FOR BIT (a row's products are added in ascending offset order in both).
Reference: indiceConv / indiceConvBackward, mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-456."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _scene(dev, batch, shape, density, seed):
    g = torch.Generator().manual_seed(seed)
    cells = batch * shape[0] * shape[1] * shape[2]
    n = max(int(cells * density), 1)
    flat = torch.randperm(cells, generator=g)[:n].sort().values
    b = flat // (shape[0] * shape[1] * shape[2])
    r = flat % (shape[0] * shape[1] * shape[2])
    z, y, x = r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]
    return torch.stack([b, z, y, x], 1).to(torch.int32).to(dev)


def _both(ops, fn):
    res = {}
    keep = ops.SORTED_CONV, ops.SPARSE_TILE_CONV
    try:
        ops.SPARSE_TILE_CONV = False
        for on in (False, True):
            ops.SORTED_CONV = on
            res[on] = fn()
    finally:
        ops.SORTED_CONV, ops.SPARSE_TILE_CONV = keep
    return res[False], res[True]


@pytest.mark.parametrize('cin,cout', [(32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (32, 128), (128, 32)])
@pytest.mark.parametrize('density', [0.03, 0.12, 0.45])
def test_sorted_rows_give_the_same_bits_as_voxel_order(dev, cin, cout, density):
    from objectcentricocccompletion_amd.spconv import ops
    shape = [20, 18, 19]
    coors = _scene(dev, 4, shape, density, seed=cin + cout)
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 4, shape, 3, subm=True)
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.1).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(n, cin, generator=g).to(dev).to(dt)
        dy = torch.randn(n, cout, generator=g).to(dev).to(dt)

        def run():
            y = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias)
            dx, _ = ops.indice_conv_backward(x, w, dy, pairs, num, False, True, need_filter_grad=False)
            return y, dx
        (y0, dx0), (y1, dx1) = _both(ops, run)
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
        assert bool(torch.isfinite(y1.float()).all()) and float(y1.float().abs().max()) > 0


@pytest.mark.parametrize('tiles', [(4, 8), (4, 4), (8, 16), (16, 16), (8, 4)])
def test_tile_plan_sizes_and_padding_rows(dev, tiles):
    """fixed-capacity padding rows (-1 coordinates: no offsets at all, not even the centre) produce the bias; every
    tile plan gives the same bits; the row count is no multiple of anything."""
    from objectcentricocccompletion_amd.spconv import ops
    shape = [17, 13, 15]
    coors = _scene(dev, 3, shape, 0.15, seed=11)
    coors = torch.cat([coors, torch.full((77, 4), -1, dtype=torch.int32, device=dev)], 0)
    n = coors.shape[0]
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(3, 3, 3, 64, 128, generator=g) * 0.1).to(dev)
    bias = torch.randn(128, generator=g).to(dev)
    x = torch.randn(n, 64, generator=g).to(dev).bfloat16()
    keep = ops.SORTED_TILES
    try:
        ops.SORTED_TILES = tiles
        _, pairs, num = ops.get_indice_pairs(coors, 3, shape, 3, subm=True)   # (a fresh rulebook: the order is cached on it)
        y0, y1 = _both(ops, lambda: ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias))
    finally:
        ops.SORTED_TILES = keep
    assert torch.equal(y0, y1)
    assert torch.equal(y1[-77:].float(), bias.bfloat16().float().expand(77, 128))


def test_row_order_is_a_permutation_grouped_by_pattern(dev):
    from objectcentricocccompletion_amd.spconv import ops
    shape = [24, 24, 24]
    coors = _scene(dev, 6, shape, 0.04, seed=3)
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 6, shape, 3, subm=True)
    rb, (table, mask, rows) = ops._tables_for(pairs, num, False, 'fwd', n, True)
    rec, hdr = ops.row_order(rb, table, rows)
    rec, hdr = rec.cpu().long(), hdr.cpu().tolist()
    perm, smask = rec[:, 0], rec[:, 1] & 0xffffffff
    assert torch.equal(perm.sort().values, torch.arange(n))
    tmask = torch.zeros(n, dtype=torch.long)
    tab = table.cpu().long()
    for k in range(27):
        tmask |= (tab[k] >= 0).long() << k
    assert torch.equal(smask, tmask[perm])
    # the record's two table entries: at the row's lowest and second lowest neighbour offsets
    for col in (2, 3):
        want = torch.full((n,), -1, dtype=torch.long)
        left = (smask & ~(1 << 13)).clone()
        for _ in range(col - 2):
            left &= left - 1                      # drop the lowest set bit
        has = left != 0
        low = (left & -left)
        kk = torch.tensor([int(v).bit_length() - 1 for v in low.tolist()])
        want[has] = tab[kk[has], perm[has]]
        assert torch.equal(rec[:, col], want)
    # classes in order 3+, 2, 1, 0 neighbours besides the centre; inside a class by the two lowest offsets
    nb = smask & ~(1 << 13)
    pc = torch.tensor([bin(int(v)).count('1') for v in nb])
    cls = 3 - pc.clamp(max=3)
    assert bool((cls[1:] >= cls[:-1]).all())
    e3, e2 = int((cls == 0).sum()), int((cls <= 1).sum())
    b_total = (n + 15) // 16
    assert hdr[0] == min((e3 + 15) // 16, b_total) and hdr[1] == min(max(hdr[0], (e2 + 15) // 16), b_total) and hdr[2] == b_total
    hb, mb = ops.SORTED_TILES
    assert hdr[3] == -(-hdr[0] // hb) - (-(hdr[1] - hdr[0]) // mb) - (-(hdr[2] - hdr[1]) // 16)
    assert hdr[4:] == [n, hb, mb, 13]
    # (the payoff: workgroup-offset iterations in this order against voxel order)
    def iters(m):
        m = torch.cat([m, torch.zeros((-len(m)) % 256, dtype=torch.long)]).view(-1, 256)
        acc = torch.zeros(m.shape[0], dtype=torch.long)
        for j in range(256):
            acc |= m[:, j]
        return sum(bin(int(v)).count('1') for v in acc)
    assert iters(smask) * 2 < iters(tmask)


def test_empty_and_tiny_tables(dev):
    from objectcentricocccompletion_amd.spconv import ops
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(3, 3, 3, 32, 64, generator=g) * 0.1).to(dev)
    for n_vox in (1, 5, 17):
        coors = _scene(dev, 1, [6, 6, 6], n_vox / 216.0, seed=n_vox)
        n = coors.shape[0]
        _, pairs, num = ops.get_indice_pairs(coors, 1, [6, 6, 6], 3, subm=True)
        x = torch.randn(n, 32, generator=g).to(dev).bfloat16()
        y0, y1 = _both(ops, lambda: ops.indice_conv(x, w, pairs, num, n, False, True))
        assert torch.equal(y0, y1)


def test_density_selects_the_order(dev):
    """unset switch: sparse rulebooks (pairs per row known and small) take the order, built with the rulebook; dense or
    unmeasured ones keep the voxel-order kernels"""
    from objectcentricocccompletion_amd.spconv import ops
    assert ops.SORTED_CONV is None and ops.SPARSE_TILE_CONV is None
    shape = [16, 16, 16]
    coors = _scene(dev, 2, shape, 0.05, seed=9)
    n = coors.shape[0]
    g = torch.Generator().manual_seed(4)
    w = (torch.randn(3, 3, 3, 64, 128, generator=g) * 0.1).to(dev)
    x = torch.randn(n, 64, generator=g).to(dev).bfloat16()
    keep = ops.DEFAULT_PAIRS_PER_ROW
    try:
        outs = []
        for ppr, sorted_expected in ((1.8, True), (9.0, False), (None, False)):
            ops.DEFAULT_PAIRS_PER_ROW = ppr
            if ppr is None:
                ops.density.reset()
            _, pairs, num = ops.get_indice_pairs(coors, 2, shape, 3, subm=True)
            rb = pairs._ococc
            assert bool(rb.orders) == sorted_expected            # built with the rulebook, or not at all
            outs.append(ops.indice_conv(x, w, pairs, num, n, False, True))
            assert bool(rb.orders) == sorted_expected
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    finally:
        ops.DEFAULT_PAIRS_PER_ROW = keep


@pytest.mark.parametrize('static', [False, True])
def test_layernorm_backward_in_the_pattern_order_dgrad(dev, static):
    """ococc_sparse_conv_sorted_lnbwd_bf16: the LayerNorm (+ GELU) backward of block L in the epilogue of block L+1's
    input-gradient pass, on the neighbour-pattern-order kernel (128 -> 64 and 64 -> 32 gathered -> written channels, the
    two LNB shapes of the configs[1] encoder).  Against (a) the same pass with the separate LN-backward launches: the
    gradients of the conv outputs are bit-identical, hence the conv weight gradients; d gamma / d beta are sums of the
    same terms grouped by other workgroups (tolerance); and (b) the tile kernel's LNB epilogue (another summation order
    inside the convolution: tolerance).  The oracle
    comparison of this kernel is test_occupancy_encoder_vs_oracle_restatement[sorted-*] / [bench-*]."""
    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
    from objectcentricocccompletion_amd.spconv import ops
    torch.manual_seed(0)
    enc = SubMOccEncoder(grouped_points=True).to(dev).train()
    xyz, feats, bidx = synthetic_object_grids(6, 1500, seed=3, device=dev)
    keep = ops.FUSE_LN_BACKWARD, ops._TILE_SHAPES, ops.DEFAULT_PAIRS_PER_ROW
    grads, outs, ran = {}, {}, {}
    # forward kernels identical in all three runs (tile kernel with LN epilogue for 32 -> 64: it leaves the link);
    # 'sorted': no tile kernel for the two dgrad shapes, so they run in pattern order
    fwd_only = {(32, 64): 2.0}
    for mode, fused, shapes in (('unfused', False, fwd_only), ('sorted', True, fwd_only), ('tile', True, dict(keep[1]))):
        ops.FUSE_LN_BACKWARD, ops._TILE_SHAPES, ops.DEFAULT_PAIRS_PER_ROW = fused, shapes, 1.755
        before = dict(ops.launches)
        try:
            for p in enc.parameters():
                p.grad = None
            out = enc(xyz, feats, bidx, 6, static=static)
            f = out.features.float()
            gen = torch.Generator(device=dev).manual_seed(5)
            (f * torch.randn(f.shape, generator=gen, device=dev)).sum().backward()
            outs[mode] = f.detach().clone()
            grads[mode] = {k: p.grad.detach().clone() for k, p in enc.named_parameters()}
        finally:
            ops.FUSE_LN_BACKWARD, ops._TILE_SHAPES, ops.DEFAULT_PAIRS_PER_ROW = keep
        ran[mode] = {k: v - before.get(k, 0) for k, v in ops.launches.items() if v - before.get(k, 0)}
    assert ran['sorted'].get('sorted_lnbwd') == 2 and 'tile_lnbwd' not in ran['sorted'], ran
    assert ran['tile'].get('tile_lnbwd') == 2 and 'sorted_lnbwd' not in ran['tile'], ran
    assert 'sorted_lnbwd' not in ran['unfused'] and ran['unfused'].get('sorted', 0) + ran['unfused'].get('sorted_ln', 0) >= 3, ran
    for mode in ('sorted', 'tile'):
        assert torch.equal(outs['unfused'], outs[mode])
        for k in grads['unfused']:
            a, b = grads['unfused'][k], grads[mode][k]
            assert bool(torch.isfinite(b).all()), (mode, k)
            if k.endswith('0.weight') and mode == 'sorted':   # conv weights: same d conv_out rows -> the same contraction
                assert torch.equal(a, b), (mode, k)
            elif k.endswith('0.weight'):
                # (the tile kernel adds a row's products centre first, the pattern-order kernel in ascending offset order:
                # its input gradients differ in the last f32 bit before the bf16 rounding)
                assert float((a - b).norm()) <= 2e-3 * float(a.norm()), (mode, k)
            else:                                # LayerNorm gamma / beta
                assert float((a - b).abs().max()) <= (1e-4 if mode == 'sorted' else 2e-3) * max(float(a.abs().max()), 1e-6), (mode, k)


@pytest.mark.parametrize('cin,cout', [(64, 128), (32, 64), (128, 64), (32, 128), (64, 32), (32, 32)])
@pytest.mark.parametrize('density', [0.04, 0.12, 0.45])
@pytest.mark.parametrize('act', [0, 1])
def test_pattern_order_kernel_layernorm_epilogue(dev, cin, cout, density, act):
    """ococc_sparse_conv_sorted_ln_bf16 (round 6: the conv -> LayerNorm -> GELU block of make_sparse_convmodule,
    sparse_block.py:216-289, in ONE launch on the neighbour-pattern-order kernel) = ococc_sparse_conv_sorted_bf16 followed
    by ococc_layernorm_act_fwd: the conv output bit for bit; the epilogue rounds to bf16 where the pair stores bf16 and
    sums in the LayerNorm kernel's order, so the statistics agree to f32 rounding and the activation except where a
    rounding tie flips.  Densities: tiles of all three classes (256-, 128- and 64-row tiles: four, two, one block per
    wave); a row count that is no multiple of anything; fixed-capacity padding rows (-1 coordinates) behind the live ones."""
    from objectcentricocccompletion_amd import _lib as L
    from objectcentricocccompletion_amd.spconv import ops
    shape = [14, 13, 11]
    coors = _scene(dev, 5, shape, density, seed=7 * cin + cout + act)
    pad = torch.full((37, 4), -1, dtype=torch.int32, device=dev)     # inert rows of the fixed-capacity form
    coors = torch.cat([coors, pad])
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 5, shape, 3, subm=True)
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.1).to(dev)
    x = torch.randn(n, cin, generator=g).to(dev).bfloat16()
    gamma = (torch.rand(cout, generator=g) + 0.5).to(dev)
    beta = (torch.rand(cout, generator=g) - 0.5).to(dev)
    keep = ops.SORTED_CONV, ops.SPARSE_TILE_CONV
    before = dict(ops.launches)
    try:
        ops.SORTED_CONV, ops.SPARSE_TILE_CONV = True, False
        conv = ops.indice_conv(x, w, pairs, num, n, False, True)
        assert ops.ln_fusion_kind(pairs, num, n, False, True, cin, cout) == 'sorted'
        fused = ops.indice_conv_ln(x, w, gamma, beta, 1e-3, act, pairs, num, n, False, True)
    finally:
        ops.SORTED_CONV, ops.SPARSE_TILE_CONV = keep
    assert ops.launches['sorted_ln'] - before.get('sorted_ln', 0) == 1 and ops.launches['sorted'] - before.get('sorted', 0) == 1
    conv_out, y, stats = fused
    assert torch.equal(conv_out, conv)
    y_ref = torch.empty_like(conv)
    stats_ref = torch.empty((n, 2), dtype=torch.float32, device=dev)
    L.check(L.lib.ococc_layernorm_act_fwd(L.ptr(conv), n, cout, L.ptr(gamma), L.ptr(beta), 1e-3, act, L.ptr(y_ref),
                                          L.ptr(stats_ref), L.BF16, L.stream()), 'ln')
    torch.cuda.synchronize()
    assert float((stats - stats_ref).abs().max()) <= 2e-6 * float(stats_ref.abs().max())
    d = (y.float() - y_ref.float()).abs()
    assert float(d.max()) <= 2e-2 * float(y_ref.float().abs().max())          # <= one bf16 step at the top value
    assert float((d > 0).float().mean()) < 2e-3                                # and only where a rounding tie flips


@pytest.mark.parametrize('cin,cout', [(64, 128), (32, 64), (128, 64)])
@pytest.mark.parametrize('act', [0, 1])
def test_pattern_order_kernel_layernorm_epilogue_vs_oracle(dev, cin, cout, act):
    """ococc_sparse_conv_sorted_ln_bf16 against the oracle's indiceConv (spconv_ops.h:300-354) followed by a float64
    LayerNorm(+GELU) on the bf16 conv output (what oracle/encoder_ref.py does for a make_sparse_convmodule block): conv
    output = RNE bf16 of the oracle's f32 result up to summation order, activation within a bf16 rounding of the float64
    value, row statistics to f32 accuracy."""
    import numpy as np
    from oracle import oracle as O
    from objectcentricocccompletion_amd.spconv import ops
    rng = np.random.default_rng(cin * 13 + cout + act)
    shape = (12, 11, 13)
    coors = _scene(dev, 3, list(shape), 0.12, seed=cin + 2 * cout)
    n = coors.shape[0]
    idx = coors.cpu().numpy()
    _, pairs, num = ops.get_indice_pairs(coors, 3, list(shape), 3, subm=True)
    ep, en = O.subm_rulebook(idx, 3, shape, (3, 3, 3))
    x = O.bf16_round(rng.standard_normal((n, cin)).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32))
    gamma = (rng.random(cout) + 0.5).astype(np.float32)
    beta = (rng.random(cout) - 0.5).astype(np.float32)
    keep = ops.SORTED_CONV, ops.SPARSE_TILE_CONV
    before = ops.launches['sorted_ln']
    try:
        ops.SORTED_CONV, ops.SPARSE_TILE_CONV = True, False
        fused = ops.indice_conv_ln(torch.from_numpy(x).to(dev).bfloat16(), torch.from_numpy(w).to(dev),
                                   torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev), 1e-3, act, pairs, num,
                                   n, False, True)
    finally:
        ops.SORTED_CONV, ops.SPARSE_TILE_CONV = keep
    assert fused is not None and ops.launches['sorted_ln'] == before + 1
    conv_out, y, stats = (t.float().cpu().numpy() for t in fused)
    ey = O.indice_conv(x, w, ep, en, n, subm=True)
    # (one bf16 step where the f32 sums differ in their last bits; sums of ~100 terms of size one that cancel to nearly
    # zero differ by the terms' f32 rounding, not by the result's)
    ulp = np.maximum(np.abs(ey), 2.0 ** -126) * 2.0 ** -7 + 1e-5
    assert (np.abs(conv_out - O.bf16_round(ey)) <= ulp).all() and (conv_out != O.bf16_round(ey)).mean() < 5e-3
    ez = O.layernorm_act(conv_out, gamma, beta, 1e-3, bool(act))
    assert (np.abs(y - ez) <= np.abs(ez) * 2.0 ** -8 + 1e-5).all()       # half an ulp of bf16
    c64 = conv_out.astype(np.float64)
    mu, rstd = c64.mean(1), 1.0 / np.sqrt(c64.var(1) + 1e-3)
    assert np.allclose(stats[:, 0], mu, rtol=1e-5, atol=1e-6) and np.allclose(stats[:, 1], rstd, rtol=1e-5)
