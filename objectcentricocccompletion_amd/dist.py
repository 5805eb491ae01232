"""Data-parallel plumbing for OcOccNet on one MI355X node: one process per GPU,
``torch.distributed`` over RCCL (backend "nccl" on ROCm), object tracklets sharded across
ranks, ONE bucketed gradient all-reduce per step -- what the reference gets from
tools/dist_train.sh:11-12 + mmcv's MMDistributedDataParallel.

xGMI is point to point (7 links x ~153 GB/s per GPU), so the 66.55 M gradients go out as a few
large flat buckets (default 32 MiB elements-aligned, bf16 on the wire optional) instead of
per-parameter messages; the two 4-byte avg-factor reductions of the loss ride in
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


class GradBuckets(object):
    """Flat gradient buckets: parameters are packed once (in reverse registration order, the
    order backward produces them) into contiguous buffers; ``all_reduce()`` copies the grads in,
    averages them across ranks with one collective per bucket and copies them back."""

    def __init__(self, params, bucket_bytes=32 << 20, wire_dtype=None):
        self.params = [p for p in params if p.requires_grad]
        self.wire_dtype = wire_dtype
        # gradients are read by pack() / all_reduce(), i.e. after backward() returned: the end-of-backward
        # parameter-gradient reductions (_deferred.py) are safe next to this exchange
        from . import _deferred
        _deferred.GRADS_READ_AFTER_BACKWARD = True
        self.buckets = []  # (flat buffer, [(param, offset, numel)])
        cur, cur_n = [], 0
        limit = max(1, bucket_bytes // 4)
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > limit:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += p.numel()
        if cur:
            self._close(cur, cur_n)

    def _close(self, items, n):
        p0 = items[0][0]
        dt = self.wire_dtype or p0.dtype
        self.buckets.append((torch.zeros(n, dtype=dt, device=p0.device), list(items)))

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

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
        self.pack()
        self.reduce()
        self.unpack()


def broadcast_parameters(module, src=0):
    """Identical initial weights on every rank (what DDP does at wrap time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
