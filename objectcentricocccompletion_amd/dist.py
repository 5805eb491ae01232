"""Data-parallel plumbing for OcOccNet on one MI355X node: one process per GPU,
``torch.distributed`` over RCCL (backend "nccl" on ROCm), object tracklets sharded across
ranks, ONE bucketed gradient all-reduce per step -- what the reference gets from
tools/dist_train.sh:11-12 + mmcv's MMDistributedDataParallel.

xGMI is point to point (7 links x ~153 GB/s per GPU), so the 66.55 M gradients go out as a few
large flat buckets (32 MiB, bf16 on the wire for the 66.55 M-parameter model, optionally launched from gradient
hooks so that they overlap the backward pass) instead of per-parameter messages; the two 4-byte avg-factor reductions of the loss ride in
losses.reduce_mean."""
import os

import torch
import torch.distributed as dist


def init_dist(backend=None):
    """Initialise from torchrun-style env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR/PORT).
    Returns (rank, world_size, local_rank).  Single process when WORLD_SIZE is unset/1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def shard_range(num_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of num_items units (tracklets) for this rank."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# gradient volume (elements) from which the wire format defaults to bf16: the 66.55 M-parameter OcOccNet sends 266 MB
# per step in f32, i.e. ~3 ms through one ~153 GB/s xGMI link in a ring; the 115 k-parameter encoder of
# configs[1] is latency bound either way and keeps f32
BF16_WIRE_FROM = 8 << 20


class GradBuckets(object):
    """Flat gradient buckets: parameters are packed once (in reverse registration order, the order backward produces
    them) into contiguous buffers and averaged across ranks with one collective per bucket.

    Two ways to drive it:
      * after the pass -- ``all_reduce()`` (= ``pack()``, ``reduce()``, ``unpack()``; the three pieces can be recorded
        into HIP graphs around the one eager RCCL call, as bench.py does);
      * ``overlap=True`` -- a post-accumulate-grad hook per parameter copies each gradient into its bucket as soon as
        autograd has produced it and launches the bucket's all-reduce when the last of its gradients has arrived, so
        the collectives of the late layers run beside the backward pass of the early ones (what the reference gets
        from MMDistributedDataParallel's reducer, apis/seq_training_apis.py:146-150); ``finish()`` after backward()
        waits for them and writes the averages back.
    ``wire_dtype``: dtype on the wire; default bf16 from 8 Mi gradient elements up, else the parameters' dtype."""

    def __init__(self, params, bucket_bytes=32 << 20, wire_dtype='auto', overlap=False):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        if wire_dtype == 'auto':
            wire_dtype = torch.bfloat16 if total >= BF16_WIRE_FROM else None
        self.wire_dtype = wire_dtype
        self.overlap = bool(overlap)
        from . import _deferred
        self._hooked = False
        self.buckets = []  # (flat buffer, [(param, offset, numel)])
        cur, cur_n = [], 0
        elem = 2 if wire_dtype in (torch.bfloat16, torch.float16) else 4
        limit = max(1, bucket_bytes // elem)
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > limit:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += p.numel()
        if cur:
            self._close(cur, cur_n)
        self._works = []
        if self.overlap:
            self._where = {}
            self._reset_pass()
            # world size 1 has nothing to exchange: no hooks, so the end-of-backward reductions stay on
            if self._active() or os.environ.get('WORLD_SIZE', '1') != '1':
                for bi, (_, items) in enumerate(self.buckets):
                    for p, off, n in items:
                        self._where[id(p)] = (bi, off, n)
                        p.register_post_accumulate_grad_hook(self._on_grad)
                self._hooked = True
        _deferred.note_gradient_reader(self, after_backward=not (self.overlap and self._hooked))

    def _close(self, items, n):
        p0 = items[0][0]
        dt = self.wire_dtype or p0.dtype
        self.buckets.append((torch.zeros(n, dtype=dt, device=p0.device), list(items)))

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    # ---- overlap mode -------------------------------------------------------------------------------------
    def _reset_pass(self):
        self._pending = [len(items) for _, items in self.buckets]
        self._got = [set() for _ in self.buckets]
        self._next = 0          # buckets [0, _next) have been handed to the collective library
        self._works = []
        self._pass_id = None    # autograd graph-task id of the pass whose gradients the buckets hold

    def _launch_ready(self):
        """Collectives go out strictly in bucket order on every rank -- bucket i only once 0..i-1 are out -- whatever
        order the gradients arrived in locally (a rank whose batch skipped a sub-module would otherwise issue its
        all-reduces in another order than its peers: a hang, or sums of mismatched buckets)."""
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._works.append(dist.all_reduce(self.buckets[self._next][0], async_op=True))
            self._next += 1

    def _on_grad(self, p):
        if not self._active():
            return
        task = torch._C._current_graph_task_id()
        if task != self._pass_id:   # first gradient of a pass: forget whatever an interrupted pass left behind
            if self._pass_id is not None and task < self._pass_id:
                # graph-task ids only grow: a LOWER one is an outer pass going on after a reentrant pass inside it
                # (torch.utils.checkpoint(use_reentrant=True), a Function that calls backward()), whose gradients
                # were taken for a new step
                raise RuntimeError('GradBuckets(overlap=True) saw gradients of a backward pass nested in another one; '
                                   'reentrant passes are not supported in overlap mode: use non-reentrant checkpointing '
                                   'or GradBuckets without overlap (pack / all_reduce after backward())')
            self._reset_pass()
            self._pass_id = task
        bi, off, n = self._where[id(p)]
        if bi < self._next or id(p) in self._got[bi]:
            return              # second accumulation into the same leaf: finish() re-packs nothing already sent
        flat = self.buckets[bi][0]
        flat[off:off + n].view_as(p).copy_(p.grad)
        self._got[bi].add(id(p))
        self._pending[bi] -= 1
        self._launch_ready()

    def finish(self):
        """Overlap mode, after backward(): pack the buckets that still wait for a gradient (parameters that took no
        part in the pass count as zero) and launch them in bucket order, wait for all collectives, write the averages
        into ``.grad``."""
        if not self._active():
            return
        try:
            for bi in range(self._next, len(self.buckets)):
                flat, items = self.buckets[bi]
                for p, off, n in items:
                    if id(p) in self._got[bi]:
                        continue
                    if p.grad is None:
                        flat[off:off + n].zero_()
                    else:
                        flat[off:off + n].view_as(p).copy_(p.grad)
                self._pending[bi] = 0
            self._launch_ready()
            for w in self._works:
                w.wait()
            self.unpack()
        finally:
            self._reset_pass()

    # ---- after-the-pass mode ------------------------------------------------------------------------------
    def pack(self):
        """Gradients -> flat buckets (one fused copy per bucket).  Device work only: may be recorded at the end
        of a forward+backward HIP graph."""
        if not self._active():
            return
        for flat, items in self.buckets:
            views = [flat[off:off + n].view_as(p) for p, off, n in items]
            have = [(v, p.grad) for v, (p, _, _) in zip(views, items) if p.grad is not None]
            for v, (p, _, _) in zip(views, items):
                if p.grad is None:
                    v.zero_()
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def reduce(self):
        """One collective per bucket, all in flight together; the only step that talks to RCCL."""
        if not self._active():
            return
        works = [dist.all_reduce(flat, async_op=True) for flat, _ in self.buckets]
        for w in works:
            w.wait()

    def unpack(self):
        """Averaged buckets -> gradients (one fused scale + copy per bucket); may open the optimizer's graph."""
        if not self._active():
            return
        inv = 1.0 / dist.get_world_size()
        for flat, items in self.buckets:
            flat.mul_(inv)
            dst, src = [], []
            for p, off, n in items:
                v = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = v.to(p.dtype).clone()
                else:
                    dst.append(p.grad)
                    src.append(v)
            if dst:
                torch._foreach_copy_(dst, src)

    def all_reduce(self):
        if self.overlap:
            return self.finish()
        self.pack()
        self.reduce()
        self.unpack()


def broadcast_parameters(module, src=0):
    """Identical initial weights on every rank (what DDP does at wrap time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
