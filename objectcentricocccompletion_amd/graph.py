"""HIP-graph capture of a launch-bound step.

The SubMConv3d encoder step is ~80 short kernels (2-90 us each); launched one by one from
Python the host cannot keep the queue full and a quarter of the step is idle gaps
(profiles/r01_g_kernel_trace).  With the fixed-capacity ("static") form of the geometry ops
(voxel.grid_unique(static=True): no device read-back, unused rows are inert) the whole
forward + backward + optimizer step has no host dependency and is recorded once into a HIP
graph; each step is then a single hipGraphLaunch.

PyTorch is used here only as the stream / graph / allocator plumbing (torch.cuda.CUDAGraph is
hipGraph on ROCm); every kernel in the graph comes in through the C ABI on the capturing
stream.  The reference has no counterpart (it launches eagerly).
"""
import os

import torch


def _join_side_streams():
    """a capture must end with every stream it forked joined again: row-order builds nobody has waited for yet"""
    from .spconv import ops as sp_ops
    sp_ops.join_pending_orders()


class GraphedStep(object):
    """Record ``fn()`` (no arguments; reads its inputs from fixed device tensors) after
    ``warmup`` eager runs on a side stream, then ``replay()`` it.

    ``fn`` must be free of host synchronisation (no .item(), no shape that depends on device
    data) and must not allocate outside torch's caching allocator.  Its return value (any
    structure of tensors) is kept and refers to graph-owned memory that each replay
    overwrites.  ``capture_ctx`` is entered around the recording only.

    No backward pass may have run on another stream before the capture: autograd pins each
    parameter's AccumulateGrad node to the stream of its first backward and would pull that
    stream into the capture (hipStreamEndCapture then crashes)."""

    def __init__(self, fn, warmup=3, pool=None, capture_ctx=None):
        import contextlib
        if os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '') != '0':
            # ROCm 7.2: with the graph packet-capture fast path on, replaying a graph of this size after
            # any new device allocation faults ("write access to a read-only page", tools/graph_bisect.py).
            raise RuntimeError('set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the HIP '
                               'runtime is loaded (before `import torch`) to use GraphedStep')
        self.fn = fn
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        from .spconv import ops as sp_ops
        self._operands = sp_ops.graph_operand_scope()   # the weight-operand buffers this capture pins (released with it)
        with (capture_ctx if capture_ctx is not None else contextlib.nullcontext()), self._operands:
            # thread_local: other threads (RCCL watchdog, autograd workers) may keep calling the runtime
            with torch.cuda.graph(self.graph, pool=pool, stream=self.stream, capture_error_mode='thread_local'):
                self.out = fn()
                _join_side_streams()
        torch.cuda.synchronize()

    def release(self):
        """let go of what the capture pinned in spconv.ops (call when the graph is discarded or about to be re-captured)"""
        ops = getattr(self, '_operands', None)
        if ops is not None:
            ops.release()

    def __del__(self):
        try:
            self.release()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass

    def pool(self):
        return self.graph.pool()

    def replay(self):
        from .spconv import ops as sp_ops
        sp_ops.refresh_graph_operands()   # (weight operands the graph reads without preparing them: see there)
        self.graph.replay()
        return self.out

    __call__ = replay


class PipelinedStep(object):
    """A training step whose input geometry is prepared one batch ahead, inside the same graph launch.

    ``prepare()`` builds everything of the NEXT batch that depends on its points only (SubMOccEncoder.geometry:
    voxelise, scatter-mean, rulebook -- a dozen short, latency-bound launches that feed nothing before the next
    step); ``train(geometry)`` is forward + backward (+ optimizer) on the CURRENT batch.  Both are recorded into one
    HIP graph, the geometry chain on a forked side stream, so it runs beside the convolutions instead of in front of
    them.  Two such graphs alternate: graph i trains on geometry buffers i and writes buffers 1-i (L.BufferPlan gives
    ``prepare`` the same caller-owned output memory every time), so a replay never writes what it reads.  The very
    first geometry is computed eagerly here.  Same results as the un-pipelined step: the data flow is unchanged,
    only the order in which independent kernels may run.

    The reference has no counterpart; its data loader prefetches batches on the host, and its voxelisation and
    rulebook run inside forward (ops/voxel/voxelize.py:10-113, ops/spconv/conv.py:146-172)."""

    def __init__(self, prepare, train, warmup=2, forward=None, tail=None):
        """``forward`` (optional): train is split as train(forward(geometry)); the fork then sits BEHIND the forward
        pass, i.e. the geometry chain runs beside the backward kernels only.
        ``tail`` (optional): the fork sits behind train(geometry) and tail() runs on the main stream beside the
        geometry chain -- for the short, low-occupancy end of a step (parameter-gradient sums, optimizer, weight
        operands of the next step), which leaves most of the chip to the geometry kernels."""
        from . import _lib as L
        self.plans = (L.BufferPlan(), L.BufferPlan())
        self.side = torch.cuda.Stream()
        self.geo = [None, None]
        for i in (0, 1):  # geometry of the first batch into set 0; set 1 records its buffers (overwritten by graph 0)
            with self.plans[i], torch.no_grad():
                self.geo[i] = prepare()
        torch.cuda.synchronize()

        def body(i):
            def fn():
                cur = torch.cuda.current_stream()
                if tail is not None:
                    out = train(self.geo[i])
                    self.side.wait_stream(cur)                  # fork behind the backward pass
                    with torch.cuda.stream(self.side), self.plans[1 - i], torch.no_grad():
                        self.geo[1 - i] = prepare()
                    tail()
                    cur.wait_stream(self.side)                  # join
                    return out
                mid = forward(self.geo[i]) if forward is not None else self.geo[i]
                self.side.wait_stream(cur)                      # fork
                with torch.cuda.stream(self.side), self.plans[1 - i], torch.no_grad():
                    self.geo[1 - i] = prepare()
                out = train(mid)
                cur.wait_stream(self.side)                      # join
                return out
            return fn

        first = GraphedStep(body(0), warmup=warmup)
        self.steps = (first, GraphedStep(body(1), warmup=warmup, pool=first.pool()))
        self.turn = 0

    def replay(self):
        out = self.steps[self.turn].replay()
        self.turn ^= 1
        return out

    __call__ = replay
