"""Host logic of the end-of-backward reduction queue (objectcentricocccompletion_amd/_deferred.py) with a CPU
stand-in for the reduce kernel: the gradient a parameter ends up with must equal the immediate one in every
autograd usage (fresh .grad, accumulation, shared parameter, autograd.grad, hooks)."""
import torch

from objectcentricocccompletion_amd import _deferred as D

_LOG = []


def _flush(jobs):
    _LOG.append(len(jobs))
    for buf, val in jobs:
        buf.copy_(val)


D.register('cpu_test', _flush)


class _Mul(torch.autograd.Function):
    """y = x * w; dw = sum_rows(g * x) -- finished late when the queue admits it."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x * w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        val = (g * x).sum(0)
        buf = torch.full((2, w.numel()), float('nan'))
        out = buf[0]
        if D.deferrable(w) and D.defer('cpu_test', (out.detach(), val), [(w, out)]):
            return g * w, None     # queued: w.grad receives it when the pass ends
        out.copy_(val)
        return g * w, out


def _setup():
    torch.manual_seed(0)
    del _LOG[:]
    return torch.nn.Parameter(torch.randn(4)), torch.randn(3, 4, requires_grad=True)


def test_fresh_grad_is_handed_out_buffer_and_one_flush_per_pass():
    w, x = _setup()
    w2 = torch.nn.Parameter(torch.randn(4))
    (_Mul.apply(_Mul.apply(x, w), w2)).sum().backward()
    assert _LOG == [2] and D.pending() == 0
    assert torch.equal(w.grad, (w2 * x).sum(0).detach())
    assert torch.equal(w2.grad, (x * w).sum(0).detach())


def test_accumulation_over_two_passes():
    w, x = _setup()
    _Mul.apply(x, w).sum().backward()
    _Mul.apply(x, w).sum().backward()  # .grad exists: the flush adds the finished sum to it
    assert _LOG == [1, 1]
    assert torch.allclose(w.grad, 2 * x.sum(0).detach())


def test_parameter_with_another_differentiable_use_in_the_same_pass():
    """loss = layer(x).sum() + an explicit L2 term on the same weight: the engine delivers the L2 part
    through AccumulateGrad, the queue adds the layer's part at the end of the pass -- nothing is lost, whichever
    comes first, with or without a gradient from an earlier pass."""
    for prior in (False, True):
        w, x = _setup()
        if prior:
            w.grad = torch.ones(4)
        (_Mul.apply(x, w).sum() + w.pow(2).sum()).backward()
        exp = x.sum(0).detach() + 2 * w.detach() + (1.0 if prior else 0.0)
        assert _LOG == [1] and torch.allclose(w.grad, exp), (prior, w.grad, exp)
        was = D.ENABLED
        try:
            D.ENABLED = False
            w2 = torch.nn.Parameter(w.detach().clone())
            (_Mul.apply(x, w2).sum() + w2.pow(2).sum()).backward()
        finally:
            D.ENABLED = was
        assert torch.allclose(w.grad, w2.grad + (1.0 if prior else 0.0))
    # a weight tied to a dense op
    w, x = _setup()
    (_Mul.apply(x, w).sum() + (x.detach() @ w).sum()).backward()
    assert torch.allclose(w.grad, 2 * x.sum(0).detach())


def test_shared_parameter_in_one_pass():
    w, x = _setup()
    _Mul.apply(_Mul.apply(x, w), w).sum().backward()  # d/dw sum(x w^2) = 2 w sum(x)
    assert torch.allclose(w.grad, (2 * w * x.sum(0)).detach())
    assert not torch.isnan(w.grad).any()


def test_autograd_grad_returns_finished_values():
    w, x = _setup()
    gw, = torch.autograd.grad(_Mul.apply(x, w).sum(), [w])
    assert torch.equal(gw, x.sum(0).detach()) and w.grad is None


def test_hooks_and_create_graph_take_the_immediate_path():
    w, x = _setup()
    seen = []
    w.register_hook(lambda g: seen.append(g.clone()))
    _Mul.apply(x, w).sum().backward()
    assert _LOG == [] and torch.equal(seen[0], x.sum(0).detach())
    w2, x2 = _setup()
    _Mul.apply(x2, w2).sum().backward(create_graph=True)
    assert _LOG == [] and torch.allclose(w2.grad, x2.sum(0))


def test_backward_with_inputs_and_unneeded_parameter():
    """backward(inputs=[w]) accumulates into w.grad like a plain backward (queued); backward(inputs=[x]) does not
    touch w at all (the layer is told so through needs_input_grad and nothing may be queued for it)."""
    w, x = _setup()
    _Mul.apply(x, w).sum().backward(inputs=[w])
    assert _LOG == [1] and torch.equal(w.grad, x.sum(0).detach()) and x.grad is None
    w, x = _setup()
    _Mul.apply(x, w).sum().backward(inputs=[x])
    assert _LOG == [] and w.grad is None and torch.equal(x.grad, w.detach().expand(3, 4))


def test_outside_backward_nothing_is_queued():
    w, _ = _setup()
    assert D.defer('cpu_test', (torch.zeros(4), torch.zeros(4)), [(w, torch.zeros(4))]) is False
    assert D.pending() == 0


def test_multi_rank_needs_the_gradient_exchange_to_opt_in(monkeypatch):
    """With > 1 rank a hook-driven exchange (torch DDP) would read the unwritten buffer: the queue stays off until
    dist.GradBuckets (which reads gradients after backward) is the exchange."""
    import torch.distributed as dist
    w, x = _setup()
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda *a: 2)
    monkeypatch.setattr(D, 'GRADS_READ_AFTER_BACKWARD', False)
    _Mul.apply(x, w).sum().backward()
    assert _LOG == [] and torch.equal(w.grad, x.sum(0).detach())
    monkeypatch.setattr(D, 'GRADS_READ_AFTER_BACKWARD', True)
    w.grad = None
    _Mul.apply(x, w).sum().backward()
    assert _LOG == [1] and torch.equal(w.grad, x.sum(0).detach())


class _Nested(torch.autograd.Function):
    """identity whose backward runs a whole backward pass of its own (what torch.utils.checkpoint(use_reentrant=True)
    does): a reentrant pass inside the running one, with its own queued reduction"""

    @staticmethod
    def forward(ctx, x, w_inner, x_inner):
        ctx.w_inner, ctx.x_inner = w_inner, x_inner
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        with torch.enable_grad():
            _Mul.apply(ctx.x_inner, ctx.w_inner).sum().backward()
        return g, None, None


def test_reentrant_pass_inside_a_pass_keeps_both_queues():
    """the inner pass has its own graph-task id; it must flush ITS queue at its end and leave the outer pass's
    queued sums alone (they used to be discarded as 'left by an interrupted pass': gradients silently lost)."""
    w, x = _setup()
    w_in = torch.nn.Parameter(torch.randn(4))
    x_in = torch.randn(5, 4)
    w_last = torch.nn.Parameter(torch.randn(4))
    # backward order: w_last (outer queue) -> nested pass (w_in: inner queue, flushed there) -> w (outer queue)
    y = _Mul.apply(_Nested.apply(_Mul.apply(x, w), w_in, x_in), w_last)
    y.sum().backward()
    assert _LOG == [1, 2] and D.pending() == 0, _LOG
    assert torch.equal(w_in.grad, x_in.sum(0))
    assert torch.equal(w_last.grad, (x * w).sum(0).detach())
    assert torch.equal(w.grad, (w_last * x).sum(0).detach())


def test_queue_of_a_pass_that_raised_is_never_run():
    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise ValueError('boom')

    w, x = _setup()
    try:   # backward order: _Mul (queues w's sum), then _Boom raises: the pass never reaches its end-of-pass callback
        _Mul.apply(_Boom.apply(x), w).sum().backward()
    except ValueError:
        pass
    assert D.pending() == 1 and w.grad is None and _LOG == []
    w3, x3 = _setup()
    _Mul.apply(x3, w3).sum().backward()
    assert D.pending() == 0 and _LOG == [1]       # only its own job ran; what the raised pass left is gone
    assert torch.equal(w3.grad, x3.sum(0)) and w.grad is None
