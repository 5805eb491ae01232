"""ococc_sparse_conv_tile_bf16 (compact-then-multiply sub-manifold convolution for sparse active sets)
against the output-stationary kernels, which are pinned to the oracle in test_gpu_spconv.py
(reference: indiceConv / indiceConvBackward, mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-456)."""
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


@pytest.mark.parametrize('cin,cout', [(32, 64), (64, 32), (128, 64), (64, 128), (32, 32), (128, 128)])
@pytest.mark.parametrize('density', [0.03, 0.45])
def test_tile_kernel_matches_output_stationary_kernels(dev, cin, cout, density):
    from objectcentricocccompletion_amd.spconv import ops
    shape = [12, 10, 11]
    coors = _scene(dev, 5, shape, density, seed=cin + cout)
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 5, shape, 3, subm=True)
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.1).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    for dt, tol in ((torch.bfloat16, 2e-2), (torch.float32, 2e-5)):
        x = torch.randn(n, cin, generator=g).to(dev).to(dt)
        dy = torch.randn(n, cout, generator=g).to(dev).to(dt)
        res = {}
        for tile in (False, True):
            ops.SPARSE_TILE_CONV = tile
            try:
                y = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias)
                dx, dw = ops.indice_conv_backward(x, w, dy, pairs, num, False, True)
                y2 = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias)
            finally:
                ops.SPARSE_TILE_CONV = None
            assert torch.equal(y, y2)                                   # deterministic, run to run
            res[tile] = (y.float(), dx.float(), dw.float())
        for a, b in zip(res[False], res[True]):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= tol * max(float(a.abs().max()), 1e-6)


def test_tile_kernel_fixed_capacity_rows_and_ragged_tail(dev):
    """Rows with -1 coordinates (fixed-capacity padding) take part in nothing and produce the bias; the row
    count is not a multiple of the tile."""
    from objectcentricocccompletion_amd.spconv import ops
    shape = [9, 9, 9]
    coors = _scene(dev, 3, shape, 0.2, seed=7)
    pad = torch.full((77, 4), -1, dtype=torch.int32, device=dev)
    coors = torch.cat([coors, pad], 0)
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 3, shape, 3, subm=True)
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(3, 3, 3, 64, 64, generator=g) * 0.1).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    x = torch.randn(n, 64, generator=g).to(dev).bfloat16()
    ops.SPARSE_TILE_CONV = True
    try:
        y = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias).float()
    finally:
        ops.SPARSE_TILE_CONV = None
    ye = ops.indice_conv(x, w, pairs, num, n, False, True, bias=bias).float()
    assert float((y - ye).abs().max()) <= 2e-2 * float(ye.abs().max())
    assert torch.equal(y[-77:], bias.bfloat16().float().expand(77, 64))


def test_density_hint_selects_the_kernel(dev):
    from objectcentricocccompletion_amd.spconv import ops
    shape = [8, 8, 8]
    coors = _scene(dev, 2, shape, 0.05, seed=11)
    _, pairs, num = ops.get_indice_pairs(coors, 2, shape, 3, subm=True)
    rb = pairs._ococc
    assert not ops._use_tile_kernel(rb, 128, 64)                # no hint: the general kernels
    ops.set_rulebook_density(pairs, 1.8)
    assert ops._use_tile_kernel(rb, 128, 64) and not ops._use_tile_kernel(rb, 16, 32)
    ops.set_rulebook_density(pairs, 2.5)
    assert ops._use_tile_kernel(rb, 128, 64) and not ops._use_tile_kernel(rb, 64, 32)
    ops.set_rulebook_density(pairs, 11.0)
    assert not ops._use_tile_kernel(rb, 128, 64)


@pytest.mark.parametrize('cin,cout', [(32, 64), (64, 32), (32, 32), (64, 64)])
@pytest.mark.parametrize('act', [0, 1])
def test_tile_kernel_layernorm_epilogue(dev, cin, cout, act):
    """ococc_sparse_conv_tile_ln_bf16 = ococc_sparse_conv_tile_bf16 followed by ococc_layernorm_act_fwd: the conv
    output bit for bit, the statistics to f32 rounding, the activated output within one bf16 step; rows past the
    last full tile and a row count that is no multiple of anything."""
    from objectcentricocccompletion_amd import _lib as L
    from objectcentricocccompletion_amd.spconv import ops
    shape = [12, 10, 11]
    coors = _scene(dev, 5, shape, 0.05, seed=3 * cin + cout)
    n = coors.shape[0]
    _, pairs, num = ops.get_indice_pairs(coors, 5, shape, 3, subm=True)
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.1).to(dev)
    x = torch.randn(n, cin, generator=g).to(dev).bfloat16()
    gamma = (torch.rand(cout, generator=g) + 0.5).to(dev)
    beta = (torch.rand(cout, generator=g) - 0.5).to(dev)
    ops.SPARSE_TILE_CONV = True
    try:
        conv = ops.indice_conv(x, w, pairs, num, n, False, True)
        fused = ops.indice_conv_ln(x, w, gamma, beta, 1e-3, act, pairs, num, n, False, True)
    finally:
        ops.SPARSE_TILE_CONV = None
    assert fused is not None
    conv_out, y, stats = fused
    assert torch.equal(conv_out, conv)
    y_ref = torch.empty_like(conv)
    stats_ref = torch.empty((n, 2), dtype=torch.float32, device=dev)
    L.check(L.lib.ococc_layernorm_act_fwd(L.ptr(conv), n, cout, L.ptr(gamma), L.ptr(beta), 1e-3, act, L.ptr(y_ref),
                                          L.ptr(stats_ref), L.BF16, L.stream()), 'ln')
    torch.cuda.synchronize()
    assert float((stats - stats_ref).abs().max()) <= 1e-4 * float(stats_ref.abs().max())
    d = (y.float() - y_ref.float()).abs()
    assert float(d.max()) <= 2e-2 * float(y_ref.float().abs().max())          # <= one bf16 step at the top value
    assert float((d > 0).float().mean()) < 0.02                                # and only where a rounding tie flips
    # refused outside its shapes
    assert L.lib.ococc_sparse_conv_tile_ln_bf16(L.ptr(x), n, 128, L.ptr(x), 27, 64, L.ptr(x), 13, n, L.ptr(gamma),
                                                L.ptr(beta), 1e-3, act, L.ptr(conv_out), L.ptr(y), L.ptr(stats),
                                                L.stream()) == -3


def test_block_fuses_layernorm_only_on_the_tile_kernel(dev):
    """make_sparse_convmodule(SubMConv3d -> LN -> GELU): with the tile kernel selected the norm runs in the conv
    epilogue (default), else as its own launch; both equal the unfused block (values and gradients)."""
    from objectcentricocccompletion_amd.sparse_block import make_sparse_convmodule
    from objectcentricocccompletion_amd.spconv import SparseConvTensor, ops
    from objectcentricocccompletion_amd.spconv import modules as spm
    torch.manual_seed(4)
    shape = [12, 10, 11]
    coors = _scene(dev, 4, shape, 0.05, seed=9)
    n = coors.shape[0]
    block = make_sparse_convmodule(32, 64, 3, 'k', padding=1, conv_type='SubMConv3d', act_type='gelu',
                                   norm_cfg=dict(type='LN', eps=1e-3)).to(dev)
    feats = torch.randn(n, 32, device=dev).bfloat16()
    dout = torch.randn(n, 64, device=dev).bfloat16()
    calls = []
    real = ops.indice_conv_ln

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)

    res = {}
    orig = (spm.FUSE_TILE_CONV_LN, spm.FUSE_CONV_LN)
    keep_sorted = ops.SORTED_CONV
    ops.indice_conv_ln = spy
    try:
        # ('stationary': the output-stationary kernels -- neither the tile kernel nor, round 6, the pattern-order kernel,
        # which has the epilogue too: 'sorted')
        for name, tile, fuse, in_order in (('tile_fused', True, True, False), ('tile_unfused', True, False, False),
                                           ('stationary', False, True, False), ('sorted', False, True, True)):
            ops.SPARSE_TILE_CONV, spm.FUSE_TILE_CONV_LN, spm.FUSE_CONV_LN, ops.SORTED_CONV = tile, fuse, False, in_order
            del calls[:]
            xin = feats.clone().requires_grad_(True)
            block.zero_grad(set_to_none=True)
            y = block(SparseConvTensor(xin, coors, shape, 4)).features
            y.backward(dout)
            torch.cuda.synchronize()
            res[name] = (len(calls), y.detach().float(), xin.grad.float(), [p.grad.float().clone() for p in block.parameters()])
    finally:
        ops.indice_conv_ln = real
        ops.SPARSE_TILE_CONV = None
        ops.SORTED_CONV = keep_sorted
        spm.FUSE_TILE_CONV_LN, spm.FUSE_CONV_LN = orig
    assert res['tile_fused'][0] == 1 and res['tile_unfused'][0] == 0 and res['stationary'][0] == 0 and res['sorted'][0] == 1
    _, yu, gxu, gpu = res['tile_unfused']
    for fused in ('tile_fused', 'sorted'):
        _, yf, gxf, gpf = res[fused]
        assert float((yf - yu).abs().max()) <= 2e-2 * float(yu.abs().max())
        assert float((gxf - gxu).abs().max()) <= 2e-2 * float(gxu.abs().max())
        for a, b in zip(gpf, gpu):
            assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-6


# ------------------------------------------------------------------ directly against the oracle
# (reference arithmetic: indiceConv / indiceConvBackward, spconv_ops.h:260-456; fp32 accumulation of exact bf16
#  products on both sides, so only the summation order differs)
ORACLE_TOL = dict(rtol=1e-4, atol=2e-4)


def _oracle_case(dev, cin, cout, density, seed):
    import numpy as np
    from oracle import oracle as O
    from objectcentricocccompletion_amd.spconv import ops
    shape = [14, 15, 16]
    coors = _scene(dev, 2, shape, density, seed=seed)
    idx = coors.cpu().numpy()
    n = len(idx)
    rng = np.random.default_rng(seed)
    x = O.bf16_round(rng.standard_normal((n, cin)).astype(np.float32))
    w = O.bf16_round(rng.standard_normal((3, 3, 3, cin, cout)).astype(np.float32) * 0.2)
    dy = O.bf16_round(rng.standard_normal((n, cout)).astype(np.float32))
    _, pairs, num = ops.get_indice_pairs(coors, 2, shape, 3, subm=True)
    ep, en = O.subm_rulebook(idx, 2, shape)
    return n, x, w, dy, pairs, num, ep, en


@pytest.mark.parametrize('cin,cout', [(32, 64), (64, 32), (128, 64), (64, 64), (32, 32), (64, 128), (128, 128)])
@pytest.mark.parametrize('density', [0.04, 0.6])
def test_tile_kernel_forward_backward_vs_oracle(dev, cin, cout, density):
    """ops.SPARSE_TILE_CONV = True routes forward and dgrad of the shapes the tile kernel covers through
    subm_tile_conv_kernel (the three launches of it in the benchmark step are <128,64,512>, <32,64,256>, <64,32,256>);
    shapes it does not cover (64x128, 128x128) fall to the streamed-weights kernel -- same bar."""
    import numpy as np
    from oracle import oracle as O
    from objectcentricocccompletion_amd.spconv import ops
    n, x, w, dy, pairs, num, ep, en = _oracle_case(dev, cin, cout, density, seed=cin * 7 + cout)
    xt, wt, dyt = (torch.from_numpy(a).to(dev) for a in (x, w, dy))
    ops.SPARSE_TILE_CONV = True
    try:
        used = ops._use_tile_kernel(pairs._ococc, cin, cout), ops._use_tile_kernel(pairs._ococc, cout, cin)
        y = ops.indice_conv(xt, wt, pairs, num, n, False, True)
        yb = ops.indice_conv(xt.bfloat16(), wt.bfloat16(), pairs, num, n, False, True)
        din, dw = ops.indice_conv_backward(xt, wt, dyt, pairs, num, False, True)
    finally:
        ops.SPARSE_TILE_CONV = None
    assert used == (cin * cout < 128 * 128, cin * cout < 128 * 128)
    ey = O.indice_conv(x, w, ep, en, n, subm=True)
    edin, edw = O.indice_conv_backward(x, w, dy, ep, en, subm=True)
    assert np.allclose(y.cpu().numpy(), ey, **ORACLE_TOL)
    assert torch.equal(yb, y.bfloat16())                       # bf16 output = RNE of the f32 accumulators
    assert np.allclose(din.cpu().numpy(), edin, **ORACLE_TOL)
    assert np.allclose(dw.cpu().numpy(), edw, rtol=1e-4, atol=2e-4 * max(1.0, float(np.abs(edw).max())))


@pytest.mark.parametrize('cin,cout', [(32, 64), (64, 32), (32, 32), (64, 64)])
@pytest.mark.parametrize('act', [0, 1])
def test_tile_kernel_layernorm_epilogue_vs_oracle(dev, cin, cout, act):
    """ococc_sparse_conv_tile_ln_bf16 against the oracle's indiceConv followed by a float64 LayerNorm(+GELU)
    (sparse_block.py:216-289 builds conv -> LN -> act; the norm reads the bf16 conv output, as the two-launch path
    and oracle/encoder_ref.py do): conv output = RNE bf16 of the oracle's f32 result up to summation order,
    activated output within a bf16 rounding of the float64 value, row statistics to f32 accuracy."""
    import numpy as np
    from oracle import oracle as O
    from objectcentricocccompletion_amd.spconv import ops
    n, x, w, dy, pairs, num, ep, en = _oracle_case(dev, cin, cout, 0.05, seed=cin * 11 + cout + act)
    rng = np.random.default_rng(5)
    gamma = (rng.random(cout) + 0.5).astype(np.float32)
    beta = (rng.random(cout) - 0.5).astype(np.float32)
    ops.SPARSE_TILE_CONV = True
    try:
        fused = ops.indice_conv_ln(torch.from_numpy(x).to(dev).bfloat16(), torch.from_numpy(w).to(dev),
                                   torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev), 1e-3, act, pairs, num,
                                   n, False, True)
    finally:
        ops.SPARSE_TILE_CONV = None
    assert fused is not None
    conv_out, y, stats = (t.float().cpu().numpy() for t in fused)
    ey = O.indice_conv(x, w, ep, en, n, subm=True)
    # bf16 of sums that differ in the last f32 bits: equal except where the sum sits on a rounding boundary
    ulp = np.maximum(np.abs(ey), 2.0 ** -126) * 2.0 ** -7
    assert (np.abs(conv_out - O.bf16_round(ey)) <= ulp).all() and (conv_out != O.bf16_round(ey)).mean() < 5e-3
    ez = O.layernorm_act(conv_out, gamma, beta, 1e-3, bool(act))
    assert (np.abs(y - ez) <= np.abs(ez) * 2.0 ** -8 + 1e-5).all()       # half an ulp of bf16
    c64 = conv_out.astype(np.float64)
    mu, rstd = c64.mean(1), 1.0 / np.sqrt(c64.var(1) + 1e-3)
    assert np.allclose(stats[:, 0], mu, rtol=1e-5, atol=1e-6) and np.allclose(stats[:, 1], rstd, rtol=1e-5)


@pytest.mark.parametrize('static', [False, True])
def test_layernorm_backward_inside_the_next_layers_dgrad(dev, static):
    """SubMOccEncoder declares its blocks a chain (functional.chain_ln_backward): the LayerNorm (+ GELU) backward of
    block L runs in the epilogue of block L+1's input-gradient kernel (ococc_sparse_conv_tile_lnbwd_bf16).  Against the
    same backward pass with the separate LN-backward launches: the conv-output gradients are bit-identical, hence the
    weight gradients too; d gamma / d beta are sums of the same terms grouped by other workgroups (tolerance)."""
    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
    from objectcentricocccompletion_amd.spconv import ops
    torch.manual_seed(0)
    enc = SubMOccEncoder(grouped_points=True).to(dev).train()
    xyz, feats, bidx = synthetic_object_grids(6, 700, seed=3, device=dev)
    grads, outs, launches = {}, {}, {}
    for fused in (False, True):
        ops.FUSE_LN_BACKWARD, ops.SPARSE_TILE_CONV = fused, True   # (the compact-then-multiply kernels for every layer)
        try:
            for p in enc.parameters():
                p.grad = None
            out = enc(xyz, feats, bidx, 6, static=static)
            f = out.features.float()
            gen = torch.Generator(device=dev).manual_seed(5)
            (f * torch.randn(f.shape, generator=gen, device=dev)).sum().backward()
            outs[fused] = f.detach().clone()
            grads[fused] = {k: p.grad.detach().clone() for k, p in enc.named_parameters()}
        finally:
            ops.FUSE_LN_BACKWARD, ops.SPARSE_TILE_CONV = True, None
    assert torch.equal(outs[False], outs[True])
    for k in grads[False]:
        a, b = grads[False][k], grads[True][k]
        assert bool(torch.isfinite(b).all()), k
        if k.endswith('0.weight'):           # conv weights: same d conv_out rows -> the same contraction
            assert torch.equal(a, b), k
        else:                                # LayerNorm gamma / beta of the three blocks
            assert float((a - b).abs().max()) <= 1e-4 * max(float(a.abs().max()), 1e-6), k
    names = [k for k in grads[True] if not k.endswith('0.weight')]
    assert len(names) == 6


def test_ln_backward_link_refuses_a_second_consumer(dev):
    """The fused form is only sound when a block's output feeds the next convolution alone; another use of the tensor
    inside a declared chain is caught in the backward pass."""
    from objectcentricocccompletion_amd import _lib as L
    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
    from objectcentricocccompletion_amd.spconv.functional import chain_ln_backward
    torch.manual_seed(0)
    enc = SubMOccEncoder(grouped_points=True).to(dev).train()
    xyz, feats, bidx = synthetic_object_grids(3, 500, seed=1, device=dev)
    from objectcentricocccompletion_amd.spconv import ops
    x = enc.geometry(xyz, feats, bidx, 3)
    ops.SPARSE_TILE_CONV = True
    try:
        with chain_ln_backward():
            a = enc.conv_layers[0](x)
            b = enc.conv_layers[1](a)
            c = enc.conv_layers[2](b)
        with pytest.raises(L.OcoccError):
            (c.features.float().sum() + b.features.float().sum()).backward()
    finally:
        ops.SPARSE_TILE_CONV = None

