"""End-of-backward parameter-gradient reductions (ococc_sparse_conv_wgrad_reduce_multi,
ococc_layernorm_param_reduce_multi via objectcentricocccompletion_amd/_deferred.py) against the immediate
per-layer reductions: bit-identical gradients in every autograd usage."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stack(dev, chans, fused):
    from objectcentricocccompletion_amd.sparse_block import make_sparse_convmodule
    from objectcentricocccompletion_amd.spconv import modules as spm
    torch.manual_seed(3)
    blocks = torch.nn.ModuleList(
        [make_sparse_convmodule(a, b, 3, 'k', padding=1, conv_type='SubMConv3d', act_type='gelu',
                                norm_cfg=dict(type='LN', eps=1e-3)) for a, b in zip(chans[:-1], chans[1:])]).to(dev)
    with torch.no_grad():
        for blk in blocks:
            blk[1].weight.uniform_(0.5, 1.5)
            blk[1].bias.uniform_(-0.5, 0.5)
    spm.FUSE_CONV_LN = fused
    return blocks


def _inputs(dev, cin, cout):
    g = torch.Generator().manual_seed(11)
    B, n = 2, 900
    cells = torch.stack([torch.randperm(1080, generator=g)[:n].sort().values + b * 1080 for b in range(B)]).flatten()
    idx = torch.stack([cells // 1080, (cells // 108) % 10, (cells // 9) % 12, cells % 9], 1).int().to(dev)
    feats = torch.randn(idx.shape[0], cin, generator=g).to(dev).bfloat16()
    dout = torch.randn(idx.shape[0], cout, generator=g).to(dev).bfloat16()
    return idx, feats, dout


def _run(blocks, idx, feats, dout, passes=1, hooks=False):
    from objectcentricocccompletion_amd.spconv import SparseConvTensor
    handles = [p.register_hook(lambda g: None) for p in blocks.parameters()] if hooks else []
    blocks.zero_grad(set_to_none=True)
    for _ in range(passes):
        t = SparseConvTensor(feats.clone().requires_grad_(True), idx, [10, 12, 9], 2)
        for blk in blocks:
            t = blk(t)
        t.features.backward(dout)
    for h in handles:
        h.remove()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in blocks.parameters()]


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('chans', [(16, 32, 64, 128), (5, 32, 48)])
def test_deferred_equals_immediate(dev, chans, fused):
    """Hooks on the parameters force the per-layer reductions; without them the whole stack is finished by one
    launch per kind at the end of the pass (padded widths like 5 -> 16 and 48 -> 64 stay immediate)."""
    from objectcentricocccompletion_amd import _deferred as D
    from objectcentricocccompletion_amd.spconv import modules as spm
    orig = spm.FUSE_CONV_LN
    try:
        blocks = _stack(dev, chans, fused)
        idx, feats, dout = _inputs(dev, chans[0], chans[-1])
        seen = []
        real = dict(D._flushers)
        for kind, fn in real.items():
            D._flushers[kind] = (lambda jobs, kind=kind, fn=fn: (seen.append((kind, len(jobs))), fn(jobs))[1])
        real_joint = dict(D._joint)
        for kinds, fn in real_joint.items():
            D._joint[kinds] = (lambda a, b, kinds=kinds, fn=fn: (seen.append((kinds, len(a), len(b))), fn(a, b))[1])
        try:
            late = _run(blocks, idx, feats, dout)
        finally:
            D._flushers.update(real)
            D._joint.update(real_joint)
        assert D.pending() == 0
        if chans == (16, 32, 64, 128):
            # ONE launch for the whole pass: the slab sums of the three convs and the d gamma / d beta sums of the
            # three LayerNorms together (ococc_backward_param_reduce_multi)
            assert seen == [(('wgrad', 'ln'), 3, 3)]
        now = _run(blocks, idx, feats, dout, hooks=True)
        for a, b in zip(late, now):
            assert torch.isfinite(a).all() and torch.equal(a, b)
    finally:
        spm.FUSE_CONV_LN = orig


def test_gradient_accumulation_over_two_passes(dev):
    from objectcentricocccompletion_amd.spconv import modules as spm
    orig = spm.FUSE_CONV_LN
    try:
        blocks = _stack(dev, (16, 32, 64), True)
        idx, feats, dout = _inputs(dev, 16, 64)
        one = _run(blocks, idx, feats, dout)
        two = _run(blocks, idx, feats, dout, passes=2)  # the second pass adds into existing .grad
        for a, b in zip(one, two):
            assert torch.equal(a + a, b)
    finally:
        spm.FUSE_CONV_LN = orig


def test_weight_with_an_explicit_regulariser_in_the_loss(dev):
    """loss = stack(x) + sum(w^2) over the conv and LayerNorm parameters.  The regulariser's part comes
    through the engine, the layers' part through the end-of-backward queue; the sum must equal the per-layer path
    (OCOCC_DEFER_PARAM_REDUCE=0)."""
    from objectcentricocccompletion_amd import _deferred as D
    from objectcentricocccompletion_amd.spconv import SparseConvTensor
    from objectcentricocccompletion_amd.spconv import modules as spm
    orig = spm.FUSE_CONV_LN
    try:
        blocks = _stack(dev, (16, 32, 64), False)
        idx, feats, dout = _inputs(dev, 16, 64)

        def run():
            blocks.zero_grad(set_to_none=True)
            t = SparseConvTensor(feats.clone().requires_grad_(True), idx, [10, 12, 9], 2)
            for blk in blocks:
                t = blk(t)
            reg = sum(p.pow(2).sum() for p in blocks.parameters())
            ((t.features.float() * dout.float()).sum() + 0.5 * reg).backward()
            torch.cuda.synchronize()
            return [p.grad.clone() for p in blocks.parameters()]

        seen = []
        real = dict(D._flushers)
        for kind, fn in real.items():
            D._flushers[kind] = (lambda jobs, kind=kind, fn=fn: (seen.append(kind), fn(jobs))[1])
        real_joint = dict(D._joint)
        for kinds, fn in real_joint.items():
            D._joint[kinds] = (lambda a, b, kinds=kinds, fn=fn: (seen.extend(kinds), fn(a, b))[1])
        try:
            late = run()
        finally:
            D._flushers.update(real)
            D._joint.update(real_joint)
        assert sorted(seen) == ['ln', 'wgrad']            # the queue was in use (both kinds, one joint launch)
        was = D.ENABLED
        try:
            D.ENABLED = False
            now = run()
        finally:
            D.ENABLED = was
        for p, a, b in zip(blocks.parameters(), late, now):
            assert torch.isfinite(a).all()
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())   # (w + dW) vs (dW + w): one rounding
            assert float((a - p.detach()).abs().max()) > 0                      # the layer's part is in there
    finally:
        spm.FUSE_CONV_LN = orig


def test_layernorm_module_shared_and_autograd_grad(dev):
    """One LayerNorm used twice in a pass (the engine sums both contributions before AccumulateGrad) and
    autograd.grad() on its parameters, against torch's f32 LayerNorm."""
    from objectcentricocccompletion_amd.norm import LayerNorm
    torch.manual_seed(5)
    ln = LayerNorm(64, eps=1e-3, fused_act='gelu').to(dev)
    ref = torch.nn.LayerNorm(64, eps=1e-3).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(ln.weight)
        ref.bias.copy_(ln.bias)
    x = torch.randn(3000, 64, device=dev)
    dy = torch.randn(3000, 64, device=dev)

    def twice(m, act):
        return act(m(act(m(x))))

    gelu = torch.nn.functional.gelu
    twice(ln, lambda t: t).backward(dy)
    twice(ref, gelu).backward(dy)
    for p, q in ((ln.weight, ref.weight), (ln.bias, ref.bias)):
        assert float((p.grad - q.grad).abs().max()) <= 1e-3 * float(q.grad.abs().max())
    gw, gb = torch.autograd.grad(ln(x), [ln.weight, ln.bias], dy)
    rw, rb = torch.autograd.grad(gelu(ref(x)), [ref.weight, ref.bias], dy)
    assert float((gw - rw).abs().max()) <= 1e-4 * float(rw.abs().max())
    assert float((gb - rb).abs().max()) <= 1e-4 * float(rb.abs().max())


def test_param_reduce_multi_c_abi(dev):
    """The C entry point by itself: partials of three layers -> the column sums, each equal to the single-layer
    launch; count 0 is a no-op and count > 16 is refused."""
    import ctypes
    from objectcentricocccompletion_amd import _lib as L
    torch.manual_seed(9)
    cs, rows = [32, 128, 24], [512, 77, 1]
    parts = [torch.randn(r, 2 * c, device=dev) for r, c in zip(rows, cs)]
    outs = [torch.empty(2, c, device=dev) for c in cs]
    vp, i32 = ctypes.c_void_p * 3, ctypes.c_int32 * 3
    L.check(L.lib.ococc_layernorm_param_reduce_multi(
        3, vp(*[p.data_ptr() for p in parts]), i32(*rows), i32(*cs), vp(*[o[0].data_ptr() for o in outs]),
        vp(*[o[1].data_ptr() for o in outs]), L.stream()), 'multi')
    torch.cuda.synchronize()
    for p, o, c in zip(parts, outs, cs):
        want = p.double().sum(0).float().reshape(2, c)
        assert float((o - want).abs().max()) <= 1e-4 * float(want.abs().max() + 1)
    assert L.lib.ococc_layernorm_param_reduce_multi(0, None, None, None, None, None, L.stream()) == 0
    assert L.lib.ococc_layernorm_param_reduce_multi(17, None, None, None, None, None, L.stream()) != 0


def test_small_weight_gradients_share_one_launch(dev):
    """the small layers' slab passes (or all three) wait for the end of the backward pass and run as ONE launch
    (ococc_sparse_conv_wgrad_multi_bf16); the gradients equal those of the per-layer launches bit for bit"""
    from objectcentricocccompletion_amd import _lib as L
    from objectcentricocccompletion_amd.spconv import ops
    blocks = _stack(dev, (16, 32, 64, 128), True)
    idx, feats, dout = _inputs(dev, 16, 128)
    calls = []
    orig = L.lib.ococc_sparse_conv_wgrad_multi_bf16

    class Spy(object):
        def __call__(self, *a):
            calls.append(1)
            return orig(*a)
    keep = ops.WGRAD_TOGETHER
    try:
        L.lib.ococc_sparse_conv_wgrad_multi_bf16 = Spy()
        got = {}
        for mode in (2, 3, 0):
            ops.WGRAD_TOGETHER = mode
            del calls[:]
            got[mode] = _run(blocks, idx, feats, dout)
            assert len(calls) == (1 if mode else 0)
    finally:
        L.lib.ococc_sparse_conv_wgrad_multi_bf16 = orig
        ops.WGRAD_TOGETHER = keep
    for mode in (2, 3):
        for a, b in zip(got[mode], got[0]):
            assert torch.equal(a, b)
