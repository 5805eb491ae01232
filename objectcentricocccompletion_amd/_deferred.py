"""Parameter-gradient reductions queued to the end of an autograd backward pass.

The weight-gradient slab sums of the sparse convolutions and the dgamma/dbeta sums of the LayerNorm layers feed
nothing before the optimizer, and each is a launch of a few dozen blocks: run per layer they cost ~5-8 us apiece
at one wave of work.  Here a layer's backward computes only its partial sums and registers the reduction; the
autograd engine's final callback finishes all of them in one launch per kind before .backward() returns (inside a
HIP-graph capture the launch simply lands behind the pass's last kernel).

The queued gradient never travels through the autograd engine: the layer's backward returns None for the
parameter, and the flush itself puts the finished tensor into ``.grad`` -- assigns it when ``.grad`` is None, adds it
otherwise.  Whatever else contributes to the same parameter in the pass (a weight-decay term written into the loss,
a weight tied to a dense op, a second use of the layer, gradients accumulated over several passes) reaches ``.grad``
through the engine's own AccumulateGrad as usual and is summed with ours, in that order; nothing ever reads a
buffer before it is written.

`deferrable` admits a parameter only when the running pass is one that accumulates into ``.grad`` for it
(`torch._C._will_engine_execute_node` on its AccumulateGrad node: True for .backward(), an error for
torch.autograd.grad(..., [p]), which wants the value handed back through the engine) and when nobody watches the
engine's hand-over: no tensor hooks, no post-accumulate hooks, no create_graph, and -- with more than one rank --
only when the gradient exchange reads ``.grad`` after backward() has returned (dist.GradBuckets does; torch DDP's
reducer hooks AccumulateGrad from C++ and would miss our contribution).  Everything else takes the immediate
per-layer reduction, which returns the gradient the ordinary way.
"""
import os

import torch

ENABLED = os.environ.get('OCOCC_DEFER_PARAM_REDUCE', '1') != '0'  # 0: always the per-layer reductions
JOINT = os.environ.get('OCOCC_DEFER_JOINT', '1') != '0'   # 0: one launch per kind of sum (profiling: which kind takes the time)
# Data-parallel wrappers that consume a gradient from a hook on its AccumulateGrad node (torch DDP's reducer: a C++
# post hook, invisible from Python) never see a gradient that bypasses the engine.  With more than one rank the
# queue therefore stays off until the code that owns the gradient exchange says it reads gradients only AFTER
# backward() has returned -- dist.GradBuckets does (pack() / all_reduce() run behind the pass).
GRADS_READ_AFTER_BACKWARD = False
_readers = {}    # id(gradient exchange object) -> it reads .grad only after backward() has returned


def note_gradient_reader(owner, after_backward):
    """A gradient exchange (dist.GradBuckets) states how it reads gradients.  The queue is on at world size > 1 only
    while EVERY live exchange reads after the pass (one hook-driven exchange turns it off for all)."""
    global GRADS_READ_AFTER_BACKWARD
    import weakref
    key = id(owner)
    _readers[key] = bool(after_backward)
    weakref.finalize(owner, _forget_reader, key)
    GRADS_READ_AFTER_BACKWARD = all(_readers.values())


def _forget_reader(key):
    global GRADS_READ_AFTER_BACKWARD
    _readers.pop(key, None)
    GRADS_READ_AFTER_BACKWARD = bool(_readers) and all(_readers.values())


_flushers = {}   # kind -> fn(list of jobs)
# One queue per backward pass, keyed by the pass's graph-task id: a reentrant pass (torch.utils.checkpoint with
# use_reentrant=True, a custom Function that calls backward()) runs INSIDE another one, has its own id and its own
# end-of-pass callback, and must neither flush nor void what the outer pass has queued.
_queues = {}     # graph-task id -> [jobs {kind: [job, ...]}, grads [(param, finished-at-flush buffer), ...], weak
                 # reference to the pass's end-of-pass callback]


_joint = {}      # (kind, kind) -> fn(jobs of the first, jobs of the second) or None when it does not apply


def register(kind, fn):
    _flushers[kind] = fn


def register_joint(kinds, fn):
    """One launch for two kinds when both have jobs queued; `fn` returns False to leave them to their own flushers."""
    _joint[tuple(kinds)] = fn


_acc_nodes = {}   # id(param) -> (weak reference to it, its AccumulateGrad node, raw stream it was found under)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _accumulate_node(p):
    """p's AccumulateGrad node.  Finding it takes a throw-away view (~10 us; a SIR layer asks for 15 parameters per
    backward), and the node stays the same object while somebody holds it: kept per parameter -- in a table of bounded
    size, since the node in turn keeps its parameter alive."""
    # (a node remembers the stream it was created under, and the engine warns -- and may synchronise -- when a kept-alive
    # node meets gradients from another stream: an entry is only good for the stream it was made on)
    sid = _raw_stream(p.device.index) if (_raw_stream is not None and p.is_cuda) else 0
    hit = _acc_nodes.get(id(p))
    if hit is not None:
        if hit[0]() is p and hit[2] == sid:
            return hit[1]
        del _acc_nodes[id(p)], hit
    with torch.enable_grad():
        node = p.view_as(p).grad_fn.next_functions[0][0]
    if len(_acc_nodes) >= 8192:
        _acc_nodes.clear()
    import weakref
    _acc_nodes[id(p)] = (weakref.ref(p), node, sid)
    return node


def _accumulates_into_grad(p):
    """True when the running backward pass will execute p's AccumulateGrad node (p.grad gets the result)."""
    node = _accumulate_node(p)
    try:
        ok = bool(torch._C._will_engine_execute_node(node))
    except RuntimeError:  # autograd.grad(..., inputs=[p]) captures the gradient instead / no pass is running
        ok = False
    if not ok:   # not a pass that uses the node (e.g. the warm-up / capture passes of a graphed callable, on their own
        _acc_nodes.pop(id(p), None)   # stream): do not keep it alive beyond this call
    return ok


def deferrable(*params):
    if not ENABLED:
        return False
    if not GRADS_READ_AFTER_BACKWARD:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
    if torch.is_grad_enabled():  # create_graph: the gradient itself must stay differentiable
        return False
    for p in params:
        if not (isinstance(p, torch.Tensor) and p.is_leaf and p.requires_grad and p.dtype == torch.float32
                and p.is_contiguous() and (p.grad is None or (p.grad.dtype == torch.float32 and p.grad.is_contiguous()))):
            return False
        if p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
            return False
        if not _accumulates_into_grad(p):
            return False
    return True


def defer(kind, job, grads):
    """Queue `job` for the kind's flusher; `grads` = [(param, buffer the job will have written), ...].  True: the
    caller's backward must return None for these parameters.  False outside a backward pass (nothing queued)."""
    q = _queue_of_running_pass()
    if q is None:
        return False
    q[0].setdefault(kind, []).append(job)
    for p, v in grads:
        q[1].append((p, v.detach()))
    return True


def defer_many(kind, jobs, grads):
    """``defer`` for several jobs of one kind at once (a whole SIR layer's LayerNorm and weight sums); ``grads`` buffers
    must be plain tensors outside any graph (they are stored as they are)."""
    q = _queue_of_running_pass()
    if q is None:
        return False
    q[0].setdefault(kind, []).extend(jobs)
    q[1].extend(grads)
    return True


class _PassEnd(object):
    """The end-of-pass callback of ONE backward pass.  The engine owns it for as long as the pass exists -- it is released
    when the pass ends, normally or by an exception -- so a dead weak reference to it says "that pass is gone"."""
    __slots__ = ('task', '__weakref__')

    def __init__(self, task):
        self.task = task

    def __call__(self):
        _flush(self.task)


def _queue_of_running_pass():
    task = torch._C._current_graph_task_id()
    q = _queues.get(task)
    if q is None:
        end = _PassEnd(task)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(end)
        except RuntimeError:  # "Final callbacks can only be installed during backward pass"
            return None
        # First job of this pass.  A backward pass that raised (OOM, a raising hook) never ran its end-of-pass callback:
        # what it queued is void -- its buffers mean nothing any more.  A pass that is still in flight (this one is a
        # reentrant pass inside it) keeps its queue.
        if not _held:
            for t in [t for t, other in _queues.items() if other[2]() is None]:
                del _queues[t]
        import weakref
        q = _queues[task] = [{}, [], weakref.ref(end)]
    return q


def pending():
    return sum(len(v) for q in _queues.values() for v in q[0].values())


def discard():
    """Drop what is queued (a backward pass that raised never runs its end-of-pass callback: its jobs must not be
    finished by a later pass, against buffers that no longer mean anything)."""
    _queues.clear()


_held = False


class hold(object):
    """Inside this context the end-of-backward callback leaves the queued reductions alone; the caller runs them with
    ``flush()`` when it wants them (graph.PipelinedStep forks the next batch's geometry in between, so that it runs
    beside the reductions and the optimizer)."""

    def __enter__(self):
        global _held
        self._prev, _held = _held, True
        return self

    def __exit__(self, *exc):
        global _held
        _held = self._prev
        if exc and exc[0] is not None:   # the pass raised under hold(): its queued sums belong to nothing
            discard()
        return False


def flush():
    """Run the queued reductions now (only needed after a backward pass under ``hold()``)."""
    global _held
    was, _held = _held, False
    try:
        for task in sorted(_queues, reverse=True):   # (inner passes first: the order their callbacks would have run in)
            _flush(task)
    finally:
        _held = was


def _flush(task=None):
    """End-of-pass callback of the running pass (``task`` None), or one held queue by its id."""
    if _held:
        return
    q = _queues.pop(torch._C._current_graph_task_id() if task is None else task, None)
    if q is None:
        return
    jobs, grads = q[0], q[1]
    for (a, b), fn in _joint.items():
        if JOINT and jobs.get(a) and jobs.get(b) and fn(jobs[a], jobs[b]):
            jobs[a], jobs[b] = [], []
    for kind, items in jobs.items():
        if items:
            _flushers[kind](items)
    for p, v in grads:
        v = v.reshape(p.shape)
        if p.grad is None:
            p.grad = v
        else:  # accumulation over passes, another differentiable use of p in this pass, or a second queued use
            p.grad.add_(v)
