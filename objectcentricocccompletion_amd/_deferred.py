"""Parameter-gradient reductions queued to the end of an autograd backward pass.

The weight-gradient slab sums of the sparse convolutions and the dgamma/dbeta sums of the LayerNorm layers feed
nothing before the optimizer, and each is a launch of a few dozen blocks: run per layer they cost ~5-8 us apiece
at one wave of work.  Here a layer's backward computes only its partial sums, hands autograd a view of the (still
unwritten) gradient buffer, and registers the reduction; the autograd engine's final callback finishes all of
them in one launch per kind before .backward() / autograd.grad() returns (inside a HIP-graph capture the launch
simply lands behind the pass's last kernel).

Handing out a buffer that is written later is only sound when nobody reads it before the flush.  `deferrable`
admits exactly the parameters for which that holds -- a single process, or dist.GradBuckets as the gradient
exchange (see GRADS_READ_AFTER_BACKWARD); an f32 leaf whose .grad is None (AccumulateGrad then keeps
the tensor it is given instead of adding to an existing one), no tensor or post-accumulate hooks, no
create_graph, not already queued in this pass (a shared weight would be summed by the engine's input buffer) --
and `_flush` re-checks the outcome: a .grad that is not the handed-out buffer (the engine cloned it) is
overwritten with the finished values.  Everything else takes the immediate per-layer reduction.

Not covered: a parameter used twice in one pass WITH a nested backward (reentrant activation checkpointing) between
its two uses -- the nested pass's flush forgets that the first use is still waiting in the engine's input buffer.
Set OCOCC_DEFER_PARAM_REDUCE=0 for such models.
"""
import os

import torch

ENABLED = os.environ.get('OCOCC_DEFER_PARAM_REDUCE', '1') != '0'  # 0: always the per-layer reductions
# Data-parallel wrappers that consume a gradient from a hook on its AccumulateGrad node (torch DDP's reducer: a C++
# post hook, invisible from Python) would copy the unwritten buffer into their bucket.  With more than one rank the
# queue therefore stays off until the code that owns the gradient exchange says it reads gradients only AFTER
# backward() has returned -- dist.GradBuckets does (pack() / all_reduce() run behind the pass).
GRADS_READ_AFTER_BACKWARD = False
_flushers = {}   # kind -> fn(list of jobs)
_jobs = {}       # kind -> [job, ...] of the running pass
_grads = []      # (param, view of the buffer handed to autograd)
_queued = set()  # id(param) of the running pass


def register(kind, fn):
    _flushers[kind] = fn


def deferrable(*params):
    if not ENABLED:
        return False
    if not GRADS_READ_AFTER_BACKWARD:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
    for p in params:
        if id(p) in _queued:
            _flush()  # second use of a parameter in one pass: finish what is pending, then reduce immediately
            return False
    if torch.is_grad_enabled():
        return False
    for p in params:
        if not (isinstance(p, torch.Tensor) and p.is_leaf and p.requires_grad and p.grad is None
                and p.dtype == torch.float32 and p.is_contiguous()):
            return False
        if p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
            return False
    return True


def defer(kind, job, grads):
    """Queue `job` for the kind's flusher; `grads` = [(param, view handed to autograd), ...].  False outside a
    backward pass (nothing queued)."""
    try:  # (one callback per job: a pass that raised never ran its callbacks, so "first of the pass" is unknowable;
        # the second and later calls of a pass find the queues empty)
        torch.autograd.Variable._execution_engine.queue_callback(_flush)
    except RuntimeError:  # "Final callbacks can only be installed during backward pass"
        return False
    _jobs.setdefault(kind, []).append(job)
    for p, v in grads:
        # (an alias, not `v` itself: a second owner of the tensor autograd is handed makes AccumulateGrad clone it)
        _grads.append((p, v.detach()))
        _queued.add(id(p))
    return True


def pending():
    return sum(len(v) for v in _jobs.values())


def _flush():
    global _jobs, _grads
    jobs, grads = _jobs, _grads
    _jobs, _grads = {}, []
    _queued.clear()
    for kind, items in jobs.items():
        if items:
            _flushers[kind](items)
    for p, v in grads:
        g = p.grad
        if g is not None and g.data_ptr() != v.data_ptr():
            g.copy_(v.reshape(g.shape))
