"""CPU suite, part 4: the N>1 path on gloo with world_size 2 -- shard ranges, bucketed
gradient all-reduce (== gradient of the concatenated batch), parameter broadcast and the
avg-factor reduce_mean of the loss.  A small torch module stands in for the HIP model (the
product ops refuse CPU tensors); the collective plumbing under test is the shipped code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from objectcentricocccompletion_amd import dist as od
    from objectcentricocccompletion_amd.losses import reduce_mean
    r, w, _ = od.init_dist('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)               # different init per rank ...
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 3))
    od.broadcast_parameters(model)              # ... made identical here
    torch.manual_seed(0)
    x_all, y_all = torch.randn(8, 6), torch.randn(8, 3)
    lo, hi = od.shard_range(8, rank, world)     # tracklets are sharded, no data-path collective
    loss = ((model(x_all[lo:hi]) - y_all[lo:hi]) ** 2).sum() / 8 * world
    loss.backward()
    buckets = od.GradBuckets(model.parameters(), bucket_bytes=256)   # forces several buckets
    assert len(buckets.buckets) > 1 and buckets.wire_dtype is None     # small model: f32 on the wire
    buckets.all_reduce()
    after_pass = [p.grad.clone() for p in model.parameters()]
    # overlap mode: the same gradients, exchanged from post-accumulate-grad hooks while backward is still running
    # (one parameter takes no part in the pass: its bucket completes in finish())
    twin = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 3))
    twin.load_state_dict(model.state_dict())
    unused = torch.nn.Parameter(torch.ones(5))
    # (buckets go out strictly in index order = reverse registration order: registered first, `unused` sits in the last)
    ob = od.GradBuckets([unused] + list(twin.parameters()), bucket_bytes=256, overlap=True)
    (((twin(x_all[lo:hi]) - y_all[lo:hi]) ** 2).sum() / 8 * world).backward()
    launched_in_pass = len(ob._works)
    ob.finish()
    overlap_equal = all(torch.allclose(a, p.grad, atol=1e-7) for a, p in zip(after_pass, twin.parameters()))
    overlap_ok = overlap_equal and launched_in_pass >= 1 and float(unused.grad.abs().sum()) == 0.0
    # bf16 on the wire (what the 66.55 M-parameter model defaults to): same averages to bf16 precision
    big = od.GradBuckets(model.parameters(), wire_dtype=torch.bfloat16)
    for p, gq in zip(model.parameters(), after_pass):
        p.grad = gq.clone()
    big.all_reduce()   # gradients already equal on both ranks: the average must reproduce them within a bf16 rounding
    bf16_ok = all(torch.allclose(p.grad, gq, rtol=1e-2, atol=1e-3) for p, gq in zip(model.parameters(), after_pass))
    for p, gq in zip(model.parameters(), after_pass):
        p.grad = gq
    avg = reduce_mean(torch.tensor([float(rank + 1)]))
    # NaiveSyncBatchNorm1d (mmdet3d/ops/norm.py:28-100): statistics over both ranks' rows
    from objectcentricocccompletion_amd.norm import NaiveSyncBatchNorm1d
    torch.manual_seed(7)
    xb = torch.randn(10, 5)                       # the same tensor on both ranks; each takes its half
    bn = NaiveSyncBatchNorm1d(5).train()
    xh = xb[rank * 5:(rank + 1) * 5].clone().requires_grad_(True)
    yb = bn(xh)
    (yb * torch.arange(5.)).sum().backward()
    # NaiveSyncBatchNorm3d (norm.py:145-198): the same over 5-D tensors
    from objectcentricocccompletion_amd.norm import NaiveSyncBatchNorm3d
    x5 = torch.randn(4, 5, 2, 3, 2)
    y3 = NaiveSyncBatchNorm3d(5).train()(x5[rank * 2:(rank + 1) * 2].clone())
    # plain numpy payloads: torch tensors travel through shared-memory handles that die with the sender
    q.put((rank, [p.grad.numpy().copy() for p in model.parameters()],
           [p.detach().numpy().copy() for p in model.parameters()], float(avg), (lo, hi),
           yb.detach().numpy().copy(), xh.grad.numpy().copy(), bn.running_mean.numpy().copy(), bool(overlap_ok),
           bool(bf16_ok), y3.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_bucketed_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, w0, a0, s0, y0, gx0, rm0, ov0, bf0, y30), (_, g1, w1, a1, s1, y1, gx1, rm1, ov1, bf1, y31) = out
    torch.manual_seed(7)
    torch.randn(10, 5)                                # (the worker's generator state: xb first, then x5)
    x5 = torch.randn(4, 5, 2, 3, 2)
    want3 = torch.nn.BatchNorm3d(5).train()(x5).detach().numpy()
    import numpy as _np
    assert _np.allclose(_np.concatenate([y30, y31], 0), want3, atol=1e-5)
    assert ov0 and ov1, 'overlap-mode gradients differ from the after-the-pass exchange'
    assert bf0 and bf1, 'bf16 wire buckets'
    g0, w0, g1, w1 = ([torch.from_numpy(a) for a in t] for t in (g0, w0, g1, w1))
    assert s0 == (0, 4) and s1 == (4, 8) and a0 == a1 == 1.5
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)                 # broadcast made the replicas identical
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b)              # both ranks hold the same averaged gradient
    # synced batch norm == plain BatchNorm1d on the concatenated batch (equal per-rank sizes)
    import numpy as np
    torch.manual_seed(7)
    xb = torch.randn(10, 5).requires_grad_(True)
    ref_bn = torch.nn.BatchNorm1d(5).train()
    yr = ref_bn(xb)
    (yr * torch.arange(5.)).sum().backward()
    assert np.allclose(np.concatenate([y0, y1]), yr.detach().numpy(), atol=1e-5)
    assert np.allclose(np.concatenate([gx0, gx1]), xb.grad.numpy(), atol=1e-5)
    assert np.allclose(rm0, rm1) and np.allclose(rm0, ref_bn.running_mean.numpy(), atol=1e-6)
    # single-process reference on the whole batch
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 3))
    with torch.no_grad():
        for p, w in zip(model.parameters(), w0):
            p.copy_(w)
    torch.manual_seed(0)
    x_all, y_all = torch.randn(8, 6), torch.randn(8, 3)
    (((model(x_all) - y_all) ** 2).sum() / 8).backward()
    for p, g in zip(model.parameters(), g0):
        assert torch.allclose(p.grad, g, atol=1e-6)


def _worker_uneven(rank, world, port, q):
    """Rank 1 skips a sub-module in its pass (heads.py skips roi_encode for an empty batch): its buckets complete only in
    finish(), yet both ranks must hand the SAME bucket sequence to the collective library."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from objectcentricocccompletion_amd import dist as od
    od.init_dist('gloo')
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    ob = od.GradBuckets(params, bucket_bytes=16 * 4, overlap=True)      # 20 elements per layer -> ~2 buckets each
    order = []
    real = dist.all_reduce

    def spy(t, *args, **kw):
        order.append([i for i, (flat, _) in enumerate(ob.buckets) if flat.data_ptr() == t.data_ptr()][0])
        return real(t, *args, **kw)
    od.dist.all_reduce = spy
    results = []
    for step in range(2):   # second pass: state of the first one must be gone
        for p in params:
            p.grad = None
        order.clear()
        x = torch.ones(3, 4) * (rank + 1 + step)
        h = a(x)
        if rank == 0:
            h = b(h)            # rank 1 never touches b
        c(h).sum().backward()
        in_pass = list(order)
        ob.finish()
        results.append((in_pass, list(order), [None if p.grad is None else p.grad.numpy().copy() for p in params]))
    od.dist.all_reduce = real
    # an interrupted pass (backward raised, finish() never ran) must not leak into the next one
    for p in params:
        p.grad = None

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')
    try:
        c(Boom.apply(a(torch.ones(3, 4)))).sum().backward()
    except RuntimeError:
        pass
    for p in params:
        p.grad = None
    c(a(torch.ones(3, 4) * (rank + 1))).sum().backward()
    ob.finish()
    after_boom = [None if p.grad is None else p.grad.numpy().copy() for p in params]
    q.put((rank, results, after_boom))
    dist.barrier()
    dist.destroy_process_group()


def test_overlap_buckets_go_out_in_index_order_when_one_rank_skips_a_module():
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, res0, boom0), (_, res1, boom1) = out
    for (in0, all0, g0), (in1, all1, g1) in zip(res0, res1):
        assert all0 == all1 == sorted(all0) and len(all0) == len(set(all0)) >= 3   # same sequence, strictly by index
        assert len(in0) >= 1                                      # rank 0 did overlap part of the exchange
        assert len(in1) < len(all1)                               # rank 1 had to wait for finish() for b's buckets
        for x, y in zip(g0, g1):
            assert np.allclose(x, y)                              # both ranks hold the same averages
    # reference for step 0: mean over ranks of the local gradients
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    c(b(a(torch.ones(3, 4)))).sum().backward()
    g_rank0 = [p.grad.clone() for m in (a, b, c) for p in m.parameters()]
    for m in (a, b, c):
        m.zero_grad()
    c(a(torch.ones(3, 4) * 2)).sum().backward()
    g_rank1 = [torch.zeros_like(p) if p.grad is None else p.grad for m in (a, b, c) for p in m.parameters()]
    for got, x, y in zip(res0[0][2], g_rank0, g_rank1):
        assert np.allclose(got, ((x + y) / 2).numpy(), atol=1e-6)
    for x, y in zip(boom0, boom1):
        assert np.allclose(x, y)


def test_shard_range_covers_everything():
    from objectcentricocccompletion_amd.dist import shard_range
    for n in (0, 1, 7, 64, 257):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_wire_dtype_defaults_to_bf16_for_the_big_model():
    from objectcentricocccompletion_amd.dist import BF16_WIRE_FROM, GradBuckets
    small = GradBuckets([torch.nn.Parameter(torch.zeros(1000))])
    big = GradBuckets([torch.nn.Parameter(torch.zeros(BF16_WIRE_FROM))])
    assert small.wire_dtype is None and big.wire_dtype == torch.bfloat16
    assert big.buckets[0][0].dtype == torch.bfloat16 and len(big.buckets) == 1
