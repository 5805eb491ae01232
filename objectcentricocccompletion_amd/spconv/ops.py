"""Rulebook + sparse-conv operators -- host mirror of mmdet3d/ops/spconv/ops.py:19-180
(get_indice_pairs, indice_conv, indice_conv_backward with the same argument order).

Device work: ococc_subm_rulebook_build, ococc_rulebook_pairs_to_table,
ococc_weight_prepare_bf16, ococc_sparse_conv_gather_gemm_bf16 (forward and dgrad),
ococc_sparse_conv_wgrad_bf16.  The reference's per-offset gather/GEMM/scatter loop
(include/spconv/spconv_ops.h:260-456) does not exist here.
"""
import collections
import contextlib
import os
import weakref

import numpy as np
import torch

from .. import _deferred
from .. import _lib as L

_KD_OK = (16, 32, 64, 128)

# launches per convolution kernel family ('tile', 'tile_ln', 'tile_lnbwd', 'sorted', 'sorted_ln', 'sorted_lnbwd', 'stationary',
# 'stationary_ln') since import: what tests, __graft_entry__.smoke() and bench.py read to state WHICH kernel produced
# the numbers they check (the choice depends on the measured rulebook density, see DensityTracker)
launches = collections.Counter()


class KernelProbe(object):
    """HIP-event timer for EVERY convolution kernel launch of a step (bench.py's roofline line reports the longest): events
    are recorded on the stream the kernel is launched on, around each launch, keyed by ``family<kd,ncols>`` (forward with
    kd = Cin / ncols = Cout, dgrad with kd = Cout / ncols = Cin; family as in ``launches``), and read back after the final
    synchronize.  repeat > 1 puts that many back-to-back launches of the (idempotent) kernel between one event pair and
    divides: the ~10 us an event pair adds around a single launch then weighs 1/repeat.  The probe never changes which
    kernel runs."""

    def __init__(self, max_events=4096, external=False, repeat=1):
        self.max_events, self.repeat = max_events, int(repeat)
        self.external = external  # events recorded while a HIP graph is being captured
        self.pairs = collections.OrderedDict()   # key -> [(start, stop)]
        self.samples = collections.OrderedDict()

    def wrap(self, key, launch):
        got = self.pairs.setdefault(key, [])
        if len(got) >= self.max_events:
            return launch()
        a, b = L.Timer(), L.Timer()  # HIP events behind the C ABI (ococc_timer_*)
        a.record(self.external)  # torch's current stream == the stream handed to the C ABI (_lib.stream())
        for _ in range(self.repeat):
            out = launch()
        b.record(self.external)
        got.append((a, b))
        return out

    def sample(self):
        """Graph mode: read the in-graph event pairs after a (synchronised) replay."""
        for key, got in self.pairs.items():
            self.samples.setdefault(key, []).extend(a.elapsed_ms(b) for a, b in got)

    def keys(self):
        return list(self.pairs)

    def count(self, key):
        return (len(self.samples.get(key, ())) if self.external else len(self.pairs.get(key, ()))) * self.repeat

    def mean_ms(self, key):
        if self.external:
            got = self.samples.get(key)
            return sum(got) / len(got) / self.repeat if got else None
        got = self.pairs.get(key)
        if not got:
            return None
        return sum(a.elapsed_ms(b) for a, b in got) / len(got) / self.repeat


_probe = None


def set_probe(probe):
    global _probe
    _probe = probe


def _launch(family, kd, ncols, fn):
    """one convolution kernel launch: counted per family, timed if a probe is set"""
    launches[family] += 1
    if _probe is not None:
        return _probe.wrap('%s<%d,%d>' % (family, kd, ncols), fn)
    return fn()


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """ops.py:19-30."""
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    """ops.py:33-43."""
    out = []
    for i in range(len(input_size)):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        out.append((input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i])
    return out


class RulebookTables(object):
    """Device-side companions of a reference-format rulebook: the offset-major gather
    tables and 16-row block masks the HIP convolution reads.  Attached to the
    indice_pairs tensor (attribute ``_ococc``) so that the reference call signature
    indice_conv(features, filters, indice_pairs, ...) keeps working unchanged."""

    def __init__(self, subm, kvol):
        self.subm = subm
        self.kvol = kvol
        self.tables = {}  # (inverse, direction) -> (table, blockmask, rows)
        self.orders = {}  # table address -> (rec, hdr, table): the neighbour-pattern row order (row_order)


def _ilist(v, ndim):
    return list(v) if isinstance(v, (list, tuple)) else [v] * ndim


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, grid=None):
    """ops.py:46-106 -> (outids, indice_pairs [K,2,N] int32, indice_pair_num [K] int32).  indice_pairs[k] is -1 past
    indice_pair_num[k] as in the reference, except for fixed-capacity coordinates (``static=True`` upstream), where
    those 27 MB of stores are skipped and the tails are unspecified."""
    L.require_device(indices)
    ndim = indices.shape[1] - 1
    ksize, stride, padding = _ilist(ksize, ndim), _ilist(stride, ndim), _ilist(padding, ndim)
    dilation, out_padding = _ilist(dilation, ndim), _ilist(out_padding, ndim)
    for d, s in zip(dilation, stride):
        assert any([s == 1, d == 1]), "don't support this."
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices = indices.contiguous()
    if ndim == 2:  # a 2-D conv is a 3-D conv on a depth-1 volume
        z = indices.new_zeros((indices.size(0), 1))
        idx3 = torch.cat([indices[:, :1], z, indices[:, 1:]], 1).contiguous()
        o, pairs, num = get_indice_pairs(idx3, batch_size, [1] + list(spatial_shape), [1] + ksize,
                                         [1] + stride, [0] + padding, [1] + dilation,
                                         [0] + out_padding, subm, transpose)
        return (indices if subm else torch.cat([o[:, :1], o[:, 2:]], 1).contiguous()), pairs, num
    if ndim != 3:
        raise NotImplementedError('only 2-D and 3-D sparse convolutions are built')
    if not subm:
        from .rulebook_regular import build_regular_rulebook
        if transpose:
            out_shape = get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation,
                                               out_padding)
        else:
            out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
        return build_regular_rulebook(indices, batch_size, out_shape, ksize, stride, padding,
                                      dilation, transpose)
    n = indices.size(0)
    kvol = int(np.prod(ksize))
    dev = indices.device
    nbytes = L.lib.ococc_subm_rulebook_workspace_bytes(n, int(batch_size), L.i3(spatial_shape),
                                                       L.i3(ksize))
    if nbytes < 0:
        raise L.OcoccError('get_indice_pairs: unsupported geometry (odd kernel sizes, '
                           'batch*D*H*W < 2^31 required)')
    ws = L.workspace(nbytes, dev)
    nbr_t = L.empty((kvol, n), torch.int32, dev)
    mask = L.empty(((n + 15) // 16,), torch.int32, dev) if kvol <= 32 else None
    pairs = L.empty((kvol, 2, n), torch.int32, dev)
    num = L.empty((kvol,), torch.int32, dev)
    grid = getattr(indices, '_ococc_grid', None)
    if (grid is not None and grid[3] == (int(batch_size),) + tuple(int(v) for v in spatial_shape)
            and all(int(d) == 1 for d in dilation)):
        # the rows come straight from grid_unique over this very grid: reuse its cell bitmap + prefix
        gws, boff, poff = grid[:3]
        # fixed-capacity coordinates (static=True upstream): nobody reads the -1 tails of the pair lists
        fill_tails = 0 if (len(grid) > 4 and grid[4]) else 1
        L.check(L.lib.ococc_subm_rulebook_build_sorted(L.ptr(indices), n, int(batch_size), L.i3(spatial_shape),
                                                       L.i3(ksize), gws.data_ptr() + boff, gws.data_ptr() + poff,
                                                       L.ptr(nbr_t), L.ptr(mask), L.ptr(pairs), L.ptr(num), fill_tails,
                                                       L.ptr(ws), ws.numel(), L.stream()),
                'subm_rulebook_build_sorted')
        pairs._ococc_keepalive = gws
    else:
        L.check(L.lib.ococc_subm_rulebook_build(L.ptr(indices), n, int(batch_size),
                                                L.i3(spatial_shape), L.i3(ksize), L.i3(dilation),
                                                L.ptr(nbr_t), L.ptr(mask), L.ptr(pairs), L.ptr(num),
                                                L.ptr(ws), ws.numel(), L.stream()),
                'subm_rulebook_build')
    dilated = any(int(d) != 1 for d in dilation)
    # (a dilated sub-manifold rulebook of the reference is not its own mirror image -- it keeps padding = k/2 --
    # so it is handled like a user-supplied one: the input-gradient table is derived from the pairs on demand)
    rb = attach_subm_tables(pairs, nbr_t, mask, n, kvol, symmetric=not dilated, num=num)
    if dilated and n > 0:
        _own_row_offset(rb, num, nbr_t, mask, n)
    return indices, pairs, num


class DensityTracker(object):
    """Rulebook pairs per output row of the sub-manifold rulebooks built recently -- what the kernel choice below keys on
    -- MEASURED ON THE DEVICE and read back without a host synchronisation: every rulebook build outside a graph capture
    queues the sum of its per-offset pair counts as an asynchronous copy into pinned memory behind an event, and the
    NEXT build (or ``poll()``) harvests it once the event has passed.  The estimate therefore lags the data by one
    build; a rulebook built before the first harvest carries no hint and runs on the output-stationary kernels (every
    kernel is correct for every table: a stale or missing estimate costs time, never results).  A captured HIP graph
    bakes in the choice made at capture time; ``regime(shape)`` tells a training loop when a re-capture would pay.

    The kernel choices key on ``stable``, not on the running average: it follows ``value`` only when that has moved by more
    than ``HYSTERESIS`` (relative) from the last committed number, so batches that jitter around a threshold of
    ``_TILE_SHAPES`` / ``SORTED_*_PAIRS_PER_ROW`` do not flip the kernel family (and with it the summation order and the
    prepared weight layouts) from step to step, whenever the copy happens to land.  One tracker per process:
    ``OCOCC_AUTO_DENSITY=0`` or ``DEFAULT_PAIRS_PER_ROW`` pin the regime for runs that must be bit-reproducible."""
    HYSTERESIS = 0.10

    def __init__(self):
        self.value = None       # pairs per row, exponentially averaged over the observed builds
        self.stable = None      # what the kernel choices see
        self.samples = 0
        self._host = None
        self._pending = None

    def observe(self, num, rows):
        if rows <= 0 or num is None or not num.is_cuda or torch.cuda.is_current_stream_capturing():
            return
        self.poll()
        if self._pending is not None:
            return                   # one observation in flight at a time
        if self._host is None:
            self._host = torch.empty((1,), dtype=torch.int64).pin_memory()
        self._host.copy_(num.sum(dtype=torch.int64).view(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (ev, int(rows))

    def poll(self, wait=False):
        """harvest the observation in flight if it has landed (``wait``: block for it -- set-up code only)"""
        if self._pending is None:
            return self.value
        ev, rows = self._pending
        if wait:
            ev.synchronize()
        if ev.query():
            v = float(self._host[0]) / rows
            self.value = v if self.value is None else 0.5 * self.value + 0.5 * v
            self.samples += 1
            self._pending = None
            if self.stable is None or abs(self.value - self.stable) > self.HYSTERESIS * self.stable:
                if self.stable is not None:
                    L.log_once(('density', round(self.stable, 1), round(self.value, 1)),
                               'rulebook density estimate moved from %.2f to %.2f pairs per row: kernel choices follow'
                               % (self.stable, self.value), level='info')
                self.stable = self.value
        return self.value

    def reset(self):
        self.value, self.stable, self.samples, self._pending = None, None, 0, None


density = DensityTracker()


def current_density():
    """the density new rulebooks are assumed to have: an explicit DEFAULT_PAIRS_PER_ROW wins, else the device-measured
    estimate (None until the first observation has landed)"""
    return float(DEFAULT_PAIRS_PER_ROW) if DEFAULT_PAIRS_PER_ROW is not None else density.stable


def attach_subm_tables(pairs, nbr_t, mask, rows, kvol, symmetric=True, num=None, rowrec=None, order=None):
    """Hang the device-side companions of a sub-manifold rulebook (offset-major gather table, 16-row block masks) on
    its indice_pairs tensor, where indice_conv / indice_conv_backward look for them.  ``num``: the per-offset pair counts
    (device), observed by the density tracker.  ``rowrec``: row records of the neighbour-pattern order, if the kernel
    that wrote the table left them (new_rulebook_wants_order); ``order``: (rec, hdr), if it left the finished order."""
    rb = RulebookTables(symmetric, kvol)
    if AUTO_DENSITY and DEFAULT_PAIRS_PER_ROW is None:
        density.observe(num, rows)
    if current_density() is not None:
        rb.pairs_per_row = current_density()
    rb.tables[(False, 'fwd')] = (nbr_t, mask, rows)
    if symmetric:
        rb.tables[(False, 'bwd')] = (nbr_t, mask, rows)  # symmetric: same table, offset-flipped weights
    pairs._ococc = rb
    if order is not None:
        rb.orders[nbr_t.data_ptr()] = [order[0], order[1], nbr_t, None]   # (nbr_t: keeps the key's address alive)
    elif rowrec is not None or (_sorted_regime(rb) and 0 < rows < ORDER_MAX_ROWS):
        # the neighbour-pattern row order is part of the geometry: built here, it runs wherever the rulebook is built
        # (bench.py: on the forked stream, beside the previous batch's convolutions) and not in front of the first layer
        # (row records at hand: the counters they were counted into must be emptied by the placing pass in any case)
        build_row_order(rb, nbr_t, rows, rowrec)
    return rb


def _tables_for(indice_pairs, indice_pair_num, inverse, direction, rows, subm):
    rb = getattr(indice_pairs, '_ococc', None)
    kvol = indice_pairs.size(0)
    if rb is None:
        rb = RulebookTables(False, kvol)  # user supplied rulebook: never assume symmetry
        indice_pairs._ococc = rb
    key = (bool(inverse), direction)
    if key not in rb.tables:
        # forward gathers from pairs[k][inverse] and writes rows pairs[k][1 - inverse]
        side = (0 if inverse else 1) if direction == 'fwd' else (1 if inverse else 0)
        dev = indice_pairs.device
        table = torch.empty((kvol, rows), dtype=torch.int32, device=dev)
        mask = torch.empty(((rows + 15) // 16,), dtype=torch.int32, device=dev) if kvol <= 32 else None
        L.check(L.lib.ococc_rulebook_pairs_to_table(L.ptr(indice_pairs), L.ptr(indice_pair_num),
                                                    kvol, indice_pairs.size(2), side, rows,
                                                    L.ptr(table), L.ptr(mask), L.stream()),
                'rulebook_pairs_to_table')
        if subm and not rb.subm and rows > 0:
            _own_row_offset(rb, indice_pair_num, table, mask, rows)
        rb.tables[key] = (table, mask, rows)
    return rb, rb.tables[key]


def _own_row_offset(rb, indice_pair_num, table, mask, rows):
    """indiceConv evaluates ONE offset of a subM layer as "own row" and never from the rulebook: the offset with
    the most pairs (indicePairMaxOffset, spconv_ops.h:273-278,300-303,320-322; backward :393-399).  For a regular
    sub-manifold rulebook that is the centre and its pairs are (j, j) anyway; for a dilated one (padding stays
    k/2) or a hand-made one it is whatever offset wins, so it is read back here -- once per rulebook, for rulebooks
    that are not our own dilation-1 ones (the reference reads indice_num back on every call)."""
    if getattr(rb, 'own_row_offset', None) is None:
        rb.own_row_offset = int(torch.argmax(indice_pair_num))  # first maximum, as std::max_element
    c = rb.own_row_offset
    table[c] = torch.arange(rows, dtype=torch.int32, device=table.device)
    if mask is not None and c < 32:
        mask |= (1 << c) if c < 31 else -(1 << 31)


def _subm_wgrad_pairs(rb, indice_pairs, indice_pair_num):
    """The pair lists the weight gradient of a subM layer contracts over when the rulebook is not one of ours with
    dilation 1: the offset with the most pairs pairs every row with itself (spconv_ops.h:393-399), whatever the
    rulebook holds."""
    hit = getattr(rb, 'wgrad_pairs', None)
    if hit is None:
        kvol, _, cap = indice_pairs.shape
        if getattr(rb, 'own_row_offset', None) is None:
            rb.own_row_offset = int(torch.argmax(indice_pair_num))
        c = rb.own_row_offset
        pairs, num = indice_pairs.clone(), indice_pair_num.clone()
        pairs[c] = torch.arange(cap, dtype=torch.int32, device=pairs.device)[None, :]
        num[c] = cap
        hit = rb.wgrad_pairs = (pairs, num)
    return hit


def _round_kd(c):
    for k in _KD_OK:
        if c <= k:
            return k
    raise NotImplementedError(f'{c} channels: the LDS-resident weight plan covers <= 128 '
                              'contraction channels per launch')


def _to_bf16_padded(t, cols):
    """[n, c] f32/bf16 -> contiguous bf16 [n, cols] (zero padded)."""
    n, c = t.shape
    if t.dtype == torch.bfloat16 and c == cols and t.is_contiguous():
        return t
    if c == cols:
        if t.dtype == torch.float32 and t.is_contiguous():
            out = torch.empty((n, cols), dtype=torch.bfloat16, device=t.device)
            L.check(L.lib.ococc_cast_f32_to_bf16(L.ptr(t), L.ptr(out), t.numel(), L.stream()), 'cast')
            return out
        return t.to(torch.bfloat16).contiguous()
    out = torch.zeros((n, cols), dtype=torch.bfloat16, device=t.device)
    out[:, :c] = t
    return out


# The weight-gradient kernels of a backward pass feed nothing before the optimizer and are each a few hundred
# latency-bound work items on a 2048-workgroup grid.  With WGRAD_AT_END the layers' backward only queues them; at the
# end of the pass they are launched TOGETHER, one per stream (a fork / join inside a captured graph), so that the three
# kernels of the encoder (20 + 14 + 10 us one after the other) share the chip; their slab sums follow in the one
# reduction launch.  MEASURED SLOWER on ROCm 7.2 and therefore off by default: 0.337 vs 0.299 ms per step -- the
# kernels do overlap (trace: 14 / 13 / 26 us side by side), but the fork costs 6-18 us before the side branches start and
# the join 11 us before the reduction does, more than the 18 us the overlap saves.  (One merged kernel was tried as
# well: the shapes' bodies inlined into one launch need 210 registers -- two workgroups per CU -- and 248 out of line.)
WGRAD_AT_END = os.environ.get('OCOCC_WGRAD_AT_END', '0') == '1'
# What does pay (round 4): the weight gradients of the encoder's layers (16 -> 32, 32 -> 64, 64 -> 128: 10 + 13 + 19 us, a
# few hundred work items each) queued to the end of the pass and run as ONE launch, ococc_sparse_conv_wgrad_multi_bf16.
# OCOCC_WGRAD_TOGETHER: 0 = each in its layer's backward, 2 = the two small shapes together (92 registers for the two
# bodies), 3 = all three (169 registers).
WGRAD_TOGETHER = int(os.environ.get('OCOCC_WGRAD_TOGETHER', '2'))
_TOGETHER_SHAPES = ((64, 128), (32, 64), (16, 32))   # (longest first)
_wgrad_streams = {}


def _wgrad_waits(kd_in, kd_out):
    return (kd_in, kd_out) in _TOGETHER_SHAPES[3 - WGRAD_TOGETHER:] if WGRAD_TOGETHER >= 2 else False


def _run_queued_wgrads(todo):
    late = [j for j in todo if len(j) > 5 and j[5] is not None]
    if not late:
        return
    if WGRAD_TOGETHER >= 2 and not WGRAD_AT_END:
        import ctypes
        shape = lambda j: (j[5][2], j[5][5])
        mine = sorted([j for j in late if _wgrad_waits(*shape(j))], key=lambda j: _TOGETHER_SHAPES.index(shape(j)))
        rest = [j for j in late if not _wgrad_waits(*shape(j))]
        while len(mine) >= 2:
            ch, mine = mine[:3], mine[3:]
            if len(mine) == 1:           # (never leave one behind)
                ch, mine = ch[:2], ch[2:] + mine
            n = len(ch)
            vp, i32, i64 = ctypes.c_void_p * n, ctypes.c_int32 * n, ctypes.c_int64 * n
            c = [j[5] for j in ch]       # (x, n_in, kd_in, dy, n_out, kd_out, pairs, num, kvol, cap)
            L.check(L.lib.ococc_sparse_conv_wgrad_multi_bf16(
                n, vp(*[q[0].data_ptr() for q in c]), vp(*[q[3].data_ptr() for q in c]), i32(*[q[2] for q in c]),
                i32(*[q[5] for q in c]), vp(*[q[6].data_ptr() for q in c]), vp(*[q[7].data_ptr() for q in c]),
                i32(*[q[8] for q in c]), i64(*[q[9] for q in c]), vp(*[j[0].data_ptr() for j in ch]),
                i64(*[j[0].numel() for j in ch]), L.stream()), 'sparse_conv_wgrad_multi')
        for job in mine + rest:
            ws, _, _, _, _, (x, n_in, kd_in, dy, n_out, kd_out, pairs, num, kvol, cap) = job
            L.check(L.lib.ococc_sparse_conv_wgrad_bf16(L.ptr(x), n_in, kd_in, L.ptr(dy), n_out, kd_out, L.ptr(pairs), L.ptr(num),
                                                       kvol, cap, None, L.ptr(ws), ws.numel(), L.stream()), 'sparse_conv_wgrad')
        return
    cur = torch.cuda.current_stream()
    dev = torch.cuda.current_device()
    sides = _wgrad_streams.setdefault(dev, [])
    while len(sides) < len(late) - 1:
        sides.append(torch.cuda.Stream())

    def launch(job):
        ws, _, _, _, _, (x, n_in, kd_in, dy, n_out, kd_out, pairs, num, kvol, cap) = job
        L.check(L.lib.ococc_sparse_conv_wgrad_bf16(L.ptr(x), n_in, kd_in, L.ptr(dy), n_out, kd_out, L.ptr(pairs), L.ptr(num),
                                                   kvol, cap, None, L.ptr(ws), ws.numel(), L.stream()), 'sparse_conv_wgrad')
    for i, job in enumerate(late[1:]):
        sides[i].wait_stream(cur)                     # fork
        with torch.cuda.stream(sides[i]):
            launch(job)
    launch(late[0])
    for i in range(len(late) - 1):
        cur.wait_stream(sides[i])                     # join (the operands stay referenced by the jobs until after it)


def _flush_wgrad_reduce(todo):
    """Finish the weight gradients whose slabs are waiting (one launch per 8 layers); jobs =
    (workspace, indice_pair_num, kvol, kd_in * kd_out, dw[, queued wgrad operands])."""
    import ctypes
    _run_queued_wgrads(todo)
    for lo in range(0, len(todo), 8):
        ch = todo[lo:lo + 8]
        n = len(ch)
        vp, i32, i64 = ctypes.c_void_p * n, ctypes.c_int32 * n, ctypes.c_int64 * n
        L.check(L.lib.ococc_sparse_conv_wgrad_reduce_multi(
            n, vp(*[c[0].data_ptr() for c in ch]), vp(*[c[1].data_ptr() for c in ch]), i32(*[c[2] for c in ch]),
            i64(*[c[3] for c in ch]), vp(*[c[4].data_ptr() for c in ch]), L.stream()), 'wgrad_reduce_multi')


_deferred.register('wgrad', _flush_wgrad_reduce)


def _flush_wgrad_and_ln(wjobs, ljobs):
    """The slab sums and the LayerNorm d gamma / d beta sums of a pass in one launch (jobs as in the two flushers)."""
    import ctypes
    if len(wjobs) > 8 or len(ljobs) > 16:
        return False
    _run_queued_wgrads(wjobs)
    n, k = len(wjobs), len(ljobs)
    vpn, i32n, i64n, vpk, i32k = ctypes.c_void_p * n, ctypes.c_int32 * n, ctypes.c_int64 * n, ctypes.c_void_p * k, ctypes.c_int32 * k
    L.check(L.lib.ococc_backward_param_reduce_multi(
        n, vpn(*[c[0].data_ptr() for c in wjobs]), vpn(*[c[1].data_ptr() for c in wjobs]), i32n(*[c[2] for c in wjobs]),
        i64n(*[c[3] for c in wjobs]), vpn(*[c[4].data_ptr() for c in wjobs]),
        k, vpk(*[j[0].data_ptr() for j in ljobs]), i32k(*[j[1] for j in ljobs]), i32k(*[j[2] for j in ljobs]),
        vpk(*[j[3][0].data_ptr() for j in ljobs]), vpk(*[j[3][1].data_ptr() for j in ljobs]), L.stream()),
        'backward_param_reduce_multi')
    return True


_deferred.register_joint(('wgrad', 'ln'), _flush_wgrad_and_ln)


class overlap_wgrad(object):
    """Context manager around a backward pass: weight gradients of the sparse convolutions are computed on a
    side stream, concurrently with the input-gradient chain; on exit the current stream waits for the side
    stream, so everything enqueued afterwards (optimizer, all-reduce) sees finished gradients.  Inside a
    HIP-graph capture this becomes a fork / join of graph branches."""
    _streams = {}

    def __enter__(self):
        global _overlap
        dev = torch.cuda.current_device()
        if dev not in overlap_wgrad._streams:
            overlap_wgrad._streams[dev] = torch.cuda.Stream()
        self.side, self.used = overlap_wgrad._streams[dev], False
        self._prev, _overlap = _overlap, self
        return self

    def __exit__(self, *exc):
        global _overlap
        _overlap = self._prev
        if self.used:
            torch.cuda.current_stream().wait_stream(self.side)
        return False


_overlap = None
_weight_cache = None  # {(data_ptr, mode, kd, nc): wn} filled by prepare_weights() for ONE forward+backward
_graph_operands = []  # operand buffers captured HIP graphs refer to (kept alive for the life of the process)
# ... and what they hold: {(data_ptr, mode, kd, nc): (wn, parameter version, weak reference to the parameter)}.  A graph
# captured on a cache hit records NO preparation launch: it relies on somebody rewriting these very buffers whenever the
# parameter changes.  They therefore stay refresh targets of the optimizer whatever later prepare_weights() calls were
# given (an evaluation forward, a second model), and graph.GraphedStep.replay re-prepares the ones a parameter write
# outside the optimizer (load_state_dict, an EMA copy) has left behind (refresh_graph_operands).
_graph_entries = {}


def prepare_weights(items):
    """Convert several conv weights in one launch.  items: [(filters, mode), ...] with mode 0 = forward
    operand, 1 = sub-manifold dgrad operand, 2 = generic dgrad operand; channel counts must already be
    kernel sizes (16/32/64/128, multiples of 16) and the tensors f32 -- other cases are left to the
    per-call path.  The results serve the conv calls of the current step (forward and backward); the next
    call replaces them, so call it once at the start of every forward."""
    import ctypes
    global _weight_cache
    previous, _weight_cache = (_weight_cache or {}), {}
    todo = []
    for filters, mode in items:
        cin, cout = filters.shape[-2], filters.shape[-1]
        if filters.dtype != torch.float32 or not filters.is_contiguous() or cin not in _KD_OK or cout not in _KD_OK:
            continue
        kd, nc = (cin, cout) if mode == 0 else (cout, cin)
        if mode in (0, 1) and current_density() is not None:
            # the layer will go through the tile kernel (same test as in indice_conv / indice_conv_backward for a
            # rulebook built under the current default density): fragment-major order
            probe_rb = RulebookTables(True, filters.numel() // (cin * cout))
            probe_rb.pairs_per_row = current_density()
            if _fragment_major(probe_rb, kd, nc):
                mode += 4
        kvol = filters.numel() // (cin * cout)
        key = (filters.data_ptr(), mode, kd, nc)
        hit = previous.get(key) or _graph_entries.get(key)
        if hit is not None and hit[2]() is filters and hit[1] == filters._version:
            _weight_cache[key] = hit    # still current: the optimizer refreshed it with the update (refresh_targets)
            if torch.cuda.is_current_stream_capturing():
                _graph_operands.append(hit[0])   # a captured graph reads and rewrites this buffer: it must outlive the cache
                _graph_entries[key] = hit
                if _capture_log is not None:
                    _capture_log.append((key, hit[0]))
            continue
        wn = torch.empty((kvol, nc, kd), dtype=torch.bfloat16, device=filters.device)
        todo.append((filters, mode, kvol, cin, cout, wn))
        _weight_cache[key] = (wn, filters._version, weakref.ref(filters))
    for lo in range(0, len(todo), 16):
        ch = todo[lo:lo + 16]
        n = len(ch)
        vp, i32 = ctypes.c_void_p * n, ctypes.c_int32 * n
        L.check(L.lib.ococc_weight_prepare_multi_bf16(
            n, vp(*[c[0].data_ptr() for c in ch]), i32(*[c[2] for c in ch]), i32(*[c[3] for c in ch]),
            i32(*[c[4] for c in ch]), i32(*[c[1] for c in ch]), vp(*[c[5].data_ptr() for c in ch]), L.stream()),
            'weight_prepare_multi')


_capture_log = None   # [(key, operand buffer)] of the capture in progress (graph.GraphedStep)


class graph_operand_scope(object):
    """Around a graph capture: remembers which operand buffers the capture pinned (``_graph_entries`` / ``_graph_operands``:
    a captured graph reads and rewrites them, so they are kept alive and kept current for it) -- ``release()`` lets go of
    them when the graph is discarded or re-captured.  Without it the tables only grew: a discarded graph's buffers stayed
    alive, stayed optimizer refresh targets and were checked on every replay."""

    def __init__(self):
        self.items = []

    def __enter__(self):
        global _capture_log
        self._prev, _capture_log = _capture_log, self.items
        return self

    def __exit__(self, *exc):
        global _capture_log
        _capture_log = self._prev
        return False

    def release(self):
        for key, wn in self.items:
            hit = _graph_entries.get(key)
            if hit is not None and hit[0] is wn:
                del _graph_entries[key]
            for i in range(len(_graph_operands) - 1, -1, -1):
                if _graph_operands[i] is wn:
                    del _graph_operands[i]
                    break
        self.items = []


def refresh_targets(param):
    """[(mode, kvol, cin, cout, wn)]: the bf16 operand layouts of ``param`` prepared for the current step.  An optimizer that
    rewrites them together with the parameter (optim.AdamW -> ococc_adamw_operands_f32) calls operands_refreshed() behind
    its launch, and the next prepare_weights() finds them current: no preparation launch in steady state."""
    out = []
    if not (_weight_cache or _graph_entries) or param.dim() < 3 or param.dtype != torch.float32 or not param.is_contiguous():
        return out
    cin, cout = param.shape[-2], param.shape[-1]
    kvol = param.numel() // (cin * cout)
    seen = set()
    for table in (_weight_cache or {}, _graph_entries):
        for (ptr, mode, kd, nc), (wn, _, ref) in table.items():
            if (ptr == param.data_ptr() and ref() is param and wn.data_ptr() not in seen
                    and (kd, nc) == ((cin, cout) if (mode & 3) == 0 else (cout, cin))):
                seen.add(wn.data_ptr())
                out.append((mode, kvol, cin, cout, wn))
    return out


def operands_refreshed(param):
    """the cached operand layouts of ``param`` hold its CURRENT values (call after the version counter was bumped)"""
    for table in (_weight_cache or {}, _graph_entries):
        for key, (wn, _, ref) in list(table.items()):
            if key[0] == param.data_ptr() and ref() is param:
                table[key] = (wn, param._version, ref)


def refresh_graph_operands():
    """Re-prepare, eagerly and in place, every operand buffer a captured graph reads whose parameter was written since
    (host check of a few version counters per replay; a launch only when something is stale)."""
    import ctypes
    stale = []
    for key, (wn, ver, ref) in list(_graph_entries.items()):
        p = ref()
        if p is None:
            del _graph_entries[key]
        elif p._version != ver:
            stale.append((key, p, wn))
    for key, p, wn in stale:
        cin, cout = p.shape[-2], p.shape[-1]
        vp, i32 = ctypes.c_void_p * 1, ctypes.c_int32 * 1
        L.check(L.lib.ococc_weight_prepare_multi_bf16(1, vp(p.data_ptr()), i32(p.numel() // (cin * cout)), i32(cin), i32(cout),
                                                      i32(key[1]), vp(wn.data_ptr()), L.stream()), 'weight_prepare_multi')
        entry = (wn, p._version, weakref.ref(p))
        _graph_entries[key] = entry
        if _weight_cache is not None and key in _weight_cache and _weight_cache[key][0] is wn:
            _weight_cache[key] = entry
    return len(stale)


def _prep_weights(filters, mode, kd_pad, nc_pad):
    """filters [..., cin, cout] -> bf16 wn [kvol, ncols, kd] for the gather-GEMM kernel."""
    if _weight_cache is not None:
        hit = _weight_cache.get((filters.data_ptr(), mode, kd_pad, nc_pad))
        # the very tensor object that was prepared (an address can be recycled by the allocator), not updated since
        if hit is not None and hit[2]() is filters and hit[1] == filters._version:
            return hit[0]
    cin, cout = filters.shape[-2], filters.shape[-1]
    w = filters.reshape(-1, cin, cout)
    kvol = w.size(0)
    cin_p, cout_p = (kd_pad, nc_pad) if (mode & 3) == 0 else (nc_pad, kd_pad)
    if (cin_p, cout_p) != (cin, cout):
        wp = torch.zeros((kvol, cin_p, cout_p), dtype=w.dtype, device=w.device)
        wp[:, :cin, :cout] = w
        w = wp
    if w.dtype not in (torch.float32, torch.bfloat16):
        w = w.float()
    w = w.contiguous()
    wn = torch.empty((kvol, nc_pad, kd_pad), dtype=torch.bfloat16, device=w.device)
    L.check(L.lib.ococc_weight_prepare_bf16(L.ptr(w), L.dtype_code(w.dtype), kvol, cin_p, cout_p,
                                            mode, L.ptr(wn), L.stream()), 'weight_prepare')
    return wn


# Sub-manifold convolutions over SPARSE active sets (a voxel has only a few neighbours, e.g. random-point
# object grids: 0.75 besides itself) run through the compact-then-multiply kernel
# (ococc_sparse_conv_tile_bf16); dense neighbourhoods (surfaces, ~10 neighbours) stay on the
# output-stationary kernels.  Both are correct for any input; this only picks the faster one.
# None = decide per rulebook from ``RulebookTables.pairs_per_row`` when the caller provided it
# (set_rulebook_density / DEFAULT_PAIRS_PER_ROW) against the per-shape thresholds of _TILE_SHAPES,
# True / False = force.
SPARSE_TILE_CONV = {'1': True, '0': False}.get(os.environ.get('OCOCC_SPARSE_TILE_CONV'))  # env: force on / off
# density assumed for rulebooks built from now on.  None (default): the device-measured estimate of ``density``
# (DensityTracker above; AUTO_DENSITY = False switches the measurement off); a number overrides it.
DEFAULT_PAIRS_PER_ROW = None
AUTO_DENSITY = os.environ.get('OCOCC_AUTO_DENSITY', '1') == '1'


def density_regime(kd, ncols):
    """'tile' / 'stationary' / None: the kernel family the CURRENT density estimate selects for a sub-manifold layer of
    this shape -- compare with what a captured graph was built under to decide on a re-capture"""
    v = current_density()
    if v is None or (kd, ncols) not in _TILE_SHAPES:
        return None
    return 'tile' if v <= _TILE_SHAPES[(kd, ncols)] else 'stationary'
# (contraction channels, columns) -> rulebook pairs per output row up to which the tile kernel measured faster
# (tools/density_sweep.py: random cells per 40^3 grid, 64 grids; 128 -> 64: 85 vs 109 us at 2.5 pairs / row, 249 vs
# 224 us at 4.1; 32 <-> 64: 31 vs 32 us at 1.8, 58 vs 48 us at 2.5)
_TILE_SHAPES = {(128, 64): 3.0, (64, 32): 2.0, (32, 64): 2.0}
# OCOCC_TILE_SHAPES="128x64:3.0,64x32:2.0" replaces the table (a shape left out never takes the tile kernel: an A/B switch
# for the pattern-order kernel on the same layer)
if os.environ.get('OCOCC_TILE_SHAPES') is not None:
    _TILE_SHAPES = {tuple(int(v) for v in item.split(':')[0].split('x')): float(item.split(':')[1])
                    for item in os.environ['OCOCC_TILE_SHAPES'].split(',') if item}


def set_rulebook_density(indice_pairs, pairs_per_row):
    """Tell the convolutions how many rulebook pairs an output row has on average (a host number the caller
    knows or measured once, e.g. before capturing a HIP graph: nothing is read back from the device here)."""
    rb = getattr(indice_pairs, '_ococc', None)
    if rb is not None:
        rb.pairs_per_row = float(pairs_per_row)


# LayerNorm backward of a block fused into the NEXT layer's dgrad (tile kernel) when the blocks form a declared chain
# (functional.chain_ln_backward; occ_encoder.SubMOccEncoder declares one)
FUSE_LN_BACKWARD = os.environ.get('OCOCC_FUSE_LN_BACKWARD', '1') == '1'


def _use_tile_kernel(rb, kd, ncols):
    if (rb is None or not rb.subm or rb.kvol % 2 == 0 or rb.kvol > 27 or kd not in (32, 64, 128) or ncols not in (32, 64, 128)
            or kd * ncols >= 128 * 128):
        return False
    if SPARSE_TILE_CONV is not None:
        return bool(SPARSE_TILE_CONV)
    ppr = getattr(rb, 'pairs_per_row', None)
    # measured (csrc/sparse_conv_tile.hip): ahead on these shapes, behind with 128 columns (256-row tiles)
    return ppr is not None and ppr <= _TILE_SHAPES.get((kd, ncols), -1.0)


# Output rows in neighbour-pattern order (csrc/sparse_conv_sorted.hip): sub-manifold layers the tile kernel does not
# take, on sparse rulebooks.  OCOCC_SORTED_CONV=0 keeps the voxel-order kernels, =1 forces the order whatever the
# density; unset: by the rulebook's pairs per row.  OCOCC_SORTED_TILES=heavy,mid sets the 16-row blocks per workgroup
# tile for rows with 3+ / 2 neighbours.
SORTED_CONV = {'1': True, '0': False}.get(os.environ.get('OCOCC_SORTED_CONV'))
SORTED_TILES = tuple(int(v) for v in os.environ.get('OCOCC_SORTED_TILES', '4,8').split(','))


# rulebook pairs per output row between which the order pays (tools/probe/sorted_density_sweep.py, 64 -> 128 forward on
# 64 grids of 40^3 cells: 27 -> 26 us at 1.2 pairs per row, 45 -> 31 at 1.8, 72 -> 48 at 2.2, 91 -> 70 at 2.5, 138 -> 136
# at 3.3, 188 -> 235 at 4.1; building the order costs ~15 us per rulebook).  Sparser: nothing to regroup; denser: most
# rows need most offsets whatever the order, and the small tiles only multiply the weight traffic.
SORTED_MIN_PAIRS_PER_ROW = float(os.environ.get('OCOCC_SORTED_MIN_PAIRS_PER_ROW', '1.5'))
SORTED_MAX_PAIRS_PER_ROW = float(os.environ.get('OCOCC_SORTED_MAX_PAIRS_PER_ROW', '3.0'))


def _sorted_regime(rb):
    if SORTED_CONV is False or rb is None or not rb.subm or rb.kvol > 32:
        return False
    if rb.orders or SORTED_CONV:   # (an order that came with the rulebook is used)
        return True
    ppr = getattr(rb, 'pairs_per_row', None)
    return ppr is not None and SORTED_MIN_PAIRS_PER_ROW <= ppr <= SORTED_MAX_PAIRS_PER_ROW


def _use_sorted_kernel(rb, kd, ncols):
    return (_sorted_regime(rb) and kd in (32, 64, 128) and ncols in (32, 64, 128) and kd * ncols < 128 * 128
            and not _use_tile_kernel(rb, kd, ncols))


_order_counters = {}   # (device, stream) -> the zero-in / zero-out counters of ococc_subm_row_order
ORDER_MAX_ROWS = 1 << 20   # a row record holds places below 2^20


def order_counters(dev, stream=None):
    """the counter buffer of the row-order builds that run on ``stream`` (default: the current one).  A build clears,
    counts into and reads its counters, and builds run on several streams at once (graph.PipelinedStep: the next
    batch's geometry on a side stream beside a lazily built order in train(); OCOCC_ORDER_SIDE_STREAM) -- so every
    stream has its own buffer: builds on one stream are ordered by the stream, builds on two never share counters."""
    stream = torch.cuda.current_stream(dev) if stream is None else stream
    key = (dev, int(stream.cuda_stream))
    hit = _order_counters.get(key)
    if hit is None:
        with torch.cuda.stream(stream):
            hit = _order_counters[key] = torch.zeros((int(L.lib.ococc_subm_row_order_counter_bytes()),), dtype=torch.uint8,
                                                     device=dev)
    busy = getattr(hit, '_ococc_busy', None)
    if busy is not None:    # a placing pass on ANOTHER stream (side-stream builds) still owns the buffer: wait for it
        stream.wait_event(busy)
        hit._ococc_busy = None
    return hit


def new_rulebook_wants_order(rows, kvol=27):
    """will a sub-manifold rulebook built NOW take the neighbour-pattern row order?  (the geometry kernel that writes the
    gather table can leave the order's row records on the way: scatter_points.object_grid_geometry asks before it
    launches)"""
    if SORTED_CONV is False or kvol > 32 or not 0 < rows < ORDER_MAX_ROWS:
        return False
    if SORTED_CONV:
        return True
    ppr = current_density()
    return ppr is not None and SORTED_MIN_PAIRS_PER_ROW <= ppr <= SORTED_MAX_PAIRS_PER_ROW


# The order is needed by the first layer that runs on it -- on the benchmark's encoder the third convolution -- and by
# nothing before, so it can be built on a side stream beside the layers in front (OCOCC_ORDER_SIDE_STREAM=1).  Off by
# default: inside a captured HIP graph the fork and join cost more than the 7 us they hide (replay span 322 us against
# 301 us with the build in line, r04e / r04d), as every two-branch graph measured on this ROCm did.
ORDER_SIDE_STREAM = os.environ.get('OCOCC_ORDER_SIDE_STREAM', '0') == '1'
_order_streams = {}
_pending_orders = []   # [event, keepalive tensors]: builds some stream may still have to wait for


def join_pending_orders():
    """make the current stream wait for every row-order build still running on the side stream (the end of a graph
    capture must not leave the side stream forked: graph.GraphedStep calls this)"""
    cur = torch.cuda.current_stream()
    for entry in _pending_orders:
        if entry[0] is not None:
            cur.wait_event(entry[0])
            entry[0] = None
            entry[1] = None
    del _pending_orders[:]


def build_row_order(rb, table, rows, rowrec=None):
    """launch the build of a sub-manifold gather table's neighbour-pattern row order (see row_order) and keep it with
    the rulebook.  ``rowrec``: the per-row records, if the kernel that wrote the table left them."""
    dev = table.device
    kvol = table.size(0)
    rec = torch.empty((max(rows, 1), 4), dtype=torch.int32, device=dev)
    hdr = torch.empty((8,), dtype=torch.int32, device=dev)
    dense_k = kvol // 2 if kvol % 2 == 1 else -1
    ws = None if rowrec is not None else L.workspace(L.lib.ococc_subm_row_order_scratch_bytes(rows), dev)
    side = None
    if ORDER_SIDE_STREAM:
        side = _order_streams.get(dev)
        if side is None:
            side = _order_streams[dev] = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream())
    # row records left by the geometry kernel were counted into the buffer of the stream THAT kernel ran on: the placing
    # pass reads (and clears) the same one; a build from the table counts on the stream it runs on
    counters = getattr(rowrec, '_ococc_counters', None) if rowrec is not None else None
    if counters is None:
        counters = order_counters(dev, side)
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        if rowrec is not None:
            L.check(L.lib.ococc_subm_row_order_place(L.ptr(rowrec), kvol, dense_k, rows, SORTED_TILES[0], SORTED_TILES[1],
                                                     L.ptr(counters), L.ptr(rec), L.ptr(hdr), L.stream()),
                    'subm_row_order_place')
        else:
            L.check(L.lib.ococc_subm_row_order(L.ptr(table), kvol, dense_k, rows, SORTED_TILES[0], SORTED_TILES[1],
                                               L.ptr(counters), L.ptr(ws), L.ptr(rec), L.ptr(hdr), L.stream()),
                    'subm_row_order')
        entry = None
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(side)
            if counters is not _order_counters.get((dev, int(side.cuda_stream))):
                counters._ococc_busy = ev   # (the stream that owns these counters waits before its next build)
            for t in (rec, hdr, rowrec, ws, table, counters):
                if t is not None:
                    t.record_stream(side)
            entry = [ev, (rowrec, ws, table)]   # (inputs stay alive until someone has waited for the build)
            _pending_orders[:] = [e for e in _pending_orders if e[0] is not None]
            _pending_orders.append(entry)
    rb.orders[table.data_ptr()] = [rec, hdr, table, entry]   # (table: keeps the key's address alive)


def row_order(rb, table, rows, rowrec=None):
    """(rec, hdr) of a sub-manifold gather table -- per slot {row, offset mask, table entries at its two lowest neighbour
    offsets} and the tile plan: built on first use unless it came with the rulebook, kept with it (every layer and
    direction that reads the table shares it).  The current stream waits for the build here."""
    hit = rb.orders.get(table.data_ptr())
    if hit is None:
        build_row_order(rb, table, rows, rowrec)
        hit = rb.orders[table.data_ptr()]
    entry = hit[3]
    if entry is not None:
        if entry[0] is not None:
            torch.cuda.current_stream().wait_event(entry[0])
            entry[0] = None
            entry[1] = None
        hit[3] = None
    return hit[0], hit[1]


def _fragment_major(rb, kd, ncols):
    """the tile kernel loads its weight fragments straight from L2 and wants them in fragment-major order (prepare
    mode + 4)"""
    return _use_tile_kernel(rb, kd, ncols)


def _gather_gemm(x_bf16, wn, table, mask, rows, bias, out_dtype, rb=None):
    kvol, ncols, kd = wn.shape
    if kvol > 32:
        # Kernel volumes above 32 offsets (5 x 5 x 5, 1 x 7 x 7, ...: the reference's indiceConv loops over any number,
        # spconv_ops.h:300-354; its configs use 3 x 3 x 3): the kernels walk at most 32 offsets -- one mask word per
        # 16-row block -- so the offsets go through in slices of 32 whose f32 partial outputs are summed.  No block
        # masks (every block of a slice is multiplied): correct, not fast.
        acc = None
        for k0 in range(0, kvol, 32):
            part = torch.empty((rows, ncols), dtype=torch.float32, device=x_bf16.device)
            L.check(L.lib.ococc_sparse_conv_gather_gemm_bf16(L.ptr(x_bf16), x_bf16.size(0), kd, L.ptr(wn[k0:k0 + 32]),
                                                             min(32, kvol - k0), ncols, L.ptr(table[k0:k0 + 32]), None, rows,
                                                             L.ptr(bias) if k0 == 0 else None, L.ptr(part),
                                                             L.dtype_code(torch.float32), L.stream()),
                    'sparse_conv_gather_gemm')
            acc = part if acc is None else acc.add_(part)
        return acc if out_dtype == torch.float32 else acc.to(out_dtype)
    out = torch.empty((rows, ncols), dtype=out_dtype, device=x_bf16.device)
    if _use_tile_kernel(rb, kd, ncols):  # (the caller prepared wn in fragment-major order under the same test)
        _launch('tile', kd, ncols, lambda: L.check(
            L.lib.ococc_sparse_conv_tile_bf16(L.ptr(x_bf16), x_bf16.size(0), kd, L.ptr(wn), kvol, ncols,
                                              L.ptr(table), kvol // 2, rows, L.ptr(bias), L.ptr(out),
                                              L.dtype_code(out_dtype), L.stream()), 'sparse_conv_tile'))
        return out
    if _use_sorted_kernel(rb, kd, ncols) and 0 < rows < ORDER_MAX_ROWS:
        rec, hdr = row_order(rb, table, rows)
        _launch('sorted', kd, ncols, lambda: L.check(
            L.lib.ococc_sparse_conv_sorted_bf16(L.ptr(x_bf16), x_bf16.size(0), kd, L.ptr(wn), kvol, ncols,
                                                L.ptr(table), L.ptr(rec), L.ptr(hdr), rows,
                                                L.ptr(bias), L.ptr(out), L.dtype_code(out_dtype), L.stream()),
            'sparse_conv_sorted'))
        return out
    _launch('stationary', kd, ncols, lambda: L.check(
        L.lib.ococc_sparse_conv_gather_gemm_bf16(L.ptr(x_bf16), x_bf16.size(0), kd, L.ptr(wn),
                                                 kvol, ncols, L.ptr(table), L.ptr(mask), rows,
                                                 L.ptr(bias), L.ptr(out),
                                                 L.dtype_code(out_dtype), L.stream()), 'sparse_conv_gather_gemm'))
    return out


# The kernels contract over at most 128 channels and write at most 128 columns per launch.  Wider layers (the
# reference's indiceConv has no limit, spconv_ops.h:260-456) run as panels: the contraction in 128-channel slices whose
# f32 partial outputs are summed, the columns in 128-wide slices side by side.
_PANEL = 128


def _panels(c):
    return [(lo, min(lo + _PANEL, c)) for lo in range(0, c, _PANEL)]


def _indice_conv_panels(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse, subm, bias, _saved):
    cin, cout = filters.shape[-2], filters.shape[-1]
    xf = features.float()
    cols = []
    for q0, q1 in _panels(cout):
        acc = None
        for p0, p1 in _panels(cin):
            part = indice_conv(xf[:, p0:p1].contiguous(), filters[..., p0:p1, q0:q1].contiguous(), indice_pairs,
                               indice_pair_num, num_activate_out, inverse, subm)
            acc = part if acc is None else acc.add_(part)
        cols.append(acc)
    out = torch.cat(cols, 1)
    if bias is not None:
        out = out + bias.float()
    if _saved is not None:
        _saved['x_bf16'] = features.to(torch.bfloat16).contiguous()
    return out.to(features.dtype)


def _indice_conv_backward_panels(features, filters, out_bp, indice_pairs, indice_pair_num, inverse, subm,
                                 need_input_grad, need_filter_grad):
    cin, cout = filters.shape[-2], filters.shape[-1]
    xf, dyf = features.float(), out_bp.float()
    din = torch.zeros_like(xf) if need_input_grad else None
    dw = torch.zeros(filters.shape, dtype=torch.float32, device=filters.device) if need_filter_grad else None
    for p0, p1 in _panels(cin):
        for q0, q1 in _panels(cout):
            gi, gw = indice_conv_backward(xf[:, p0:p1].contiguous(), filters[..., p0:p1, q0:q1].contiguous(),
                                          dyf[:, q0:q1].contiguous(), indice_pairs, indice_pair_num, inverse, subm,
                                          need_input_grad=need_input_grad, need_filter_grad=need_filter_grad)
            if need_input_grad:
                din[:, p0:p1] += gi
            if need_filter_grad:
                dw[..., p0:p1, q0:q1] = gw
    return (din.to(features.dtype) if need_input_grad else None), (dw.to(filters.dtype) if need_filter_grad else None)


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out,
                inverse=False, subm=False, bias=None, _saved=None):
    """ops.py:109-125.  features [n_in,Cin] (f32 or bf16), filters [kD,kH,kW,Cin,Cout];
    returns [num_activate_out, Cout] in features.dtype (f32 accumulation either way)."""
    L.require_device(features, filters, indice_pairs)
    cin, cout = filters.shape[-2], filters.shape[-1]
    assert features.shape[1] == cin
    if cin > _PANEL or cout > _PANEL:
        return _indice_conv_panels(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse, subm,
                                   bias, _saved)
    rb, (table, mask, rows) = _tables_for(indice_pairs, indice_pair_num, inverse, 'fwd',
                                          int(num_activate_out), subm)
    kd = _round_kd(cin)
    nc = (cout + 15) // 16 * 16
    x = _to_bf16_padded(features, kd)
    tile = _fragment_major(rb if subm else None, kd, nc)
    wn = _prep_weights(filters, 4 if tile else 0, kd, nc)  # +4: fragment-major order for the tile kernel
    b = None
    if bias is not None:
        b = torch.zeros((nc,), dtype=torch.float32, device=features.device)
        b[:cout] = bias.float()
    out_dtype = torch.bfloat16 if features.dtype == torch.bfloat16 else torch.float32
    out = _gather_gemm(x, wn, table, mask, rows, b, out_dtype, rb if subm else None)
    if _saved is not None:
        _saved['x_bf16'] = x
    return out if nc == cout else out[:, :cout].contiguous()


SORTED_CONV_LN = os.environ.get('OCOCC_SORTED_CONV_LN', '1') == '1'


def _sorted_ln_shape(rb, cin, cout, rows):
    return _use_sorted_kernel(rb, cin, cout) and 0 < rows < ORDER_MAX_ROWS


def _tile_ln_shape(cin, cout):
    return cin in (32, 64) and cout in (32, 64)  # instantiations of ococc_sparse_conv_tile_ln_bf16


def ln_fusion_kind(indice_pairs, indice_pair_num, num_activate_out, inverse, subm, cin, cout):
    """Which kernel a conv -> LayerNorm(+GELU) block would fuse into: 'tile' (the compact-then-multiply kernel has
    the finished f32 row in LDS: the epilogue is nearly free and the separate LN launch disappears), 'first' (the
    16 -> 32 input layer, measured ahead fused) or 'stationary' (the other output-stationary kernels: measured no
    faster than the two launches, opt-in)."""
    if (cin, cout) == (16, 32):
        return 'first'  # the resident-weights kernel at two 16-row blocks per wave: 19.9 us against 15.5 + 7.4 us
    if not subm:
        return 'stationary'
    rb, _ = _tables_for(indice_pairs, indice_pair_num, inverse, 'fwd', int(num_activate_out), subm)
    if _tile_ln_shape(cin, cout) and _use_tile_kernel(rb, cin, cout):
        return 'tile'
    # 'sorted' (round 6): rows in neighbour-pattern order, the finished row in the registers of four lanes -- the
    # 64 -> 128 forward of configs[1], whose separate LN launch read 32 MB straight back
    if SORTED_CONV_LN and _sorted_ln_shape(rb, cin, cout, int(num_activate_out)):
        return 'sorted'
    return 'stationary'


def indice_conv_ln(features, filters, gamma, beta, eps, act, indice_pairs, indice_pair_num, num_activate_out,
                   inverse=False, subm=False, _saved=None):
    """indice_conv with the LayerNorm(+GELU) that follows it in make_sparse_convmodule fused into the
    kernel epilogue.  Returns (conv_out, y, mean_rstd), or None when the shape has no fused kernel
    (caller then runs the two ops separately).  bf16 features only."""
    cin, cout = filters.shape[-2], filters.shape[-1]
    if features.dtype != torch.bfloat16 or cin not in _KD_OK or cout % 16 != 0:
        return None
    L.require_device(features, filters, indice_pairs)
    rb, (table, mask, rows) = _tables_for(indice_pairs, indice_pair_num, inverse, 'fwd',
                                          int(num_activate_out), subm)
    x = _to_bf16_padded(features, cin)
    tile = subm and _tile_ln_shape(cin, cout) and _use_tile_kernel(rb, cin, cout)
    in_order = (not tile) and subm and SORTED_CONV_LN and _sorted_ln_shape(rb, cin, cout, rows)
    wn = _prep_weights(filters, 4 if tile else 0, cin, cout)
    conv_out = torch.empty((rows, cout), dtype=torch.bfloat16, device=x.device)
    y = torch.empty_like(conv_out)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
    kvol = wn.shape[0]
    if tile:
        def run():
            return L.lib.ococc_sparse_conv_tile_ln_bf16(L.ptr(x), x.size(0), cin, L.ptr(wn), kvol, cout, L.ptr(table),
                                                        kvol // 2, rows, L.ptr(g32), L.ptr(b32), float(eps), int(act),
                                                        L.ptr(conv_out), L.ptr(y), L.ptr(stats), L.stream())
    elif in_order:
        rec, hdr = row_order(rb, table, rows)

        def run():
            return L.lib.ococc_sparse_conv_sorted_ln_bf16(L.ptr(x), x.size(0), cin, L.ptr(wn), kvol, cout, L.ptr(table),
                                                          L.ptr(rec), L.ptr(hdr), rows, L.ptr(g32), L.ptr(b32), float(eps),
                                                          int(act), L.ptr(conv_out), L.ptr(y), L.ptr(stats), L.stream())
    else:
        def run():
            return L.lib.ococc_sparse_conv_gather_gemm_ln_bf16(L.ptr(x), x.size(0), cin, L.ptr(wn), kvol, cout,
                                                               L.ptr(table), L.ptr(mask), rows, L.ptr(g32), L.ptr(b32),
                                                               float(eps), int(act), L.ptr(conv_out), L.ptr(y),
                                                               L.ptr(stats), L.stream())
    family = 'tile_ln' if tile else ('sorted_ln' if in_order else 'stationary_ln')
    rc = _launch(family, cin, cout, run)
    if rc == -3:  # OCOCC_EUNSUPPORTED: no fused kernel for this shape
        launches[family] -= 1
        return None
    L.check(rc, 'sparse_conv_gather_gemm_ln')
    if _saved is not None:
        _saved['x_bf16'] = x
        _saved['g32'], _saved['b32'] = g32, b32
    return conv_out, y, stats


def fused_indice_conv(features, filters, bias, indice_pairs, indice_pair_num, num_activate_out,
                      inverse, subm):
    """ops.py:128-139: convolution with the bias folded into the output (here: added in the
    kernel epilogue, from registers)."""
    return indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse,
                       subm, bias=bias)


def indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out):
    """ops.py:162-172 of the reference (indice_maxpool_fp32 / _half): out[o] = max(0, features[i] over the pairs (i, o)) --
    the reference's output starts at zero (pool_ops.h:34, src/maxpool.cc:9-27).  float32 or bfloat16 (its half)."""
    L.require_device(features, indice_pairs)
    if features.dtype == torch.float16:   # (the reference's _half instantiation: the maximum is exact in any format)
        return indice_maxpool(features.float(), indice_pairs, indice_pair_num, num_activate_out).half()
    if features.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError
    x = features.contiguous()
    rb, (table, _, rows) = _tables_for(indice_pairs, indice_pair_num, False, 'fwd', int(num_activate_out), False)
    out = torch.empty((int(num_activate_out), x.size(1)), dtype=x.dtype, device=x.device)
    L.check(L.lib.ococc_indice_maxpool(L.ptr(x), L.dtype_code(x.dtype), x.size(0), x.size(1), L.ptr(table), table.size(0),
                                       int(num_activate_out), L.ptr(out), L.stream()), 'indice_maxpool')
    return out


def indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num):
    """ops.py:175-184: input_bp[i] += out_bp[o] for every pair (i, o) with features[i] == out_features[o]
    (src/maxpool.cc:31-53), offsets in ascending order."""
    L.require_device(features, out_features, out_bp, indice_pairs)
    if features.dtype == torch.float16:   # (f16 values are exact in f32: the equality test selects the same rows)
        return indice_maxpool_backward(features.float(), out_features.float(), out_bp.float(), indice_pairs,
                                       indice_pair_num).half()
    if features.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError
    x, y, dy = features.contiguous(), out_features.contiguous(), out_bp.to(features.dtype).contiguous()
    rb, (table, _, rows) = _tables_for(indice_pairs, indice_pair_num, False, 'bwd', x.size(0), False)
    din = torch.empty_like(x)
    L.check(L.lib.ococc_indice_maxpool_backward(L.ptr(x), L.ptr(y), L.ptr(dy), L.dtype_code(x.dtype), x.size(0), x.size(1),
                                                L.ptr(table), table.size(0), y.size(0), L.ptr(din), L.stream()),
            'indice_maxpool_backward')
    return din


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False,
                         subm=False, _x_bf16=None, need_input_grad=True, need_filter_grad=True, _autograd=False,
                         _ln_link=None):
    """ops.py:142-160 -> (input_bp [n_in,Cin], filters_bp like filters).  ``_autograd`` (set by our autograd
    Functions only): the weight gradient may join the end-of-backward reduction queue (_deferred), in which case
    filters_bp is None here and ``filters.grad`` receives it when the pass ends."""
    L.require_device(features, filters, out_bp, indice_pairs)
    cin, cout = filters.shape[-2], filters.shape[-1]
    if cin > _PANEL or cout > _PANEL:
        return _indice_conv_backward_panels(features, filters, out_bp, indice_pairs, indice_pair_num, inverse, subm,
                                            need_input_grad, need_filter_grad)
    n_in, n_out = features.size(0), out_bp.size(0)
    kd_in, kd_out = _round_kd(cin), _round_kd(cout)
    dy = _to_bf16_padded(out_bp, kd_out)
    input_bp = filters_bp = None
    # (weight gradient first: with overlap_wgrad() it forks onto the side stream before the long dgrad kernel)
    if need_filter_grad:
        x = _x_bf16 if _x_bf16 is not None else _to_bf16_padded(features, kd_in)
        kvol = indice_pairs.size(0)
        cap = indice_pairs.size(2)
        pairs = indice_pairs
        rb0 = getattr(indice_pairs, '_ococc', None)
        if subm and cap == n_in and cap > 0 and (rb0 is None or not rb0.subm):
            if rb0 is None:
                rb0 = indice_pairs._ococc = RulebookTables(False, kvol)
            pairs, indice_pair_num = _subm_wgrad_pairs(rb0, indice_pairs, indice_pair_num)
        if inverse:  # wgrad pairs x rows with dy rows: swap the two pair rows
            pairs = pairs.flip(1).contiguous()

        def run():
            nbytes = L.lib.ococc_sparse_conv_wgrad_workspace_bytes(kvol, cap, kd_in, kd_out)
            ws = L.workspace(nbytes, features.device)
            dw = torch.empty((kvol, kd_in, kd_out), dtype=torch.float32, device=features.device)
            out = dw[:, :cin, :cout].reshape(filters.shape)  # a view: splitting kvol never copies
            # Inside an autograd backward pass only the slabs are computed now; the slab reductions of all layers
            # go into ONE launch queued to the end of the pass (_deferred): dW feeds nothing before that.
            # the slabs themselves wait for the end of the pass as well: all layers (WGRAD_AT_END), or the small shapes
            # that then share one launch
            late = _autograd and (WGRAD_AT_END or _wgrad_waits(kd_in, kd_out))
            compute = (x, n_in, kd_in, dy, n_out, kd_out, pairs, indice_pair_num, kvol, cap) if late else None
            defer = (_autograd and _overlap is None and cap > 0 and n_in > 0 and n_out > 0 and cin == kd_in
                     and cout == kd_out and _deferred.deferrable(filters)
                     and _deferred.defer('wgrad', (ws, indice_pair_num, int(kvol), kd_in * kd_out, dw, compute),
                                         [(filters, out)]))
            if defer and late:
                return None
            L.check(L.lib.ococc_sparse_conv_wgrad_bf16(L.ptr(x), n_in, kd_in, L.ptr(dy), n_out, kd_out,
                                                       L.ptr(pairs), L.ptr(indice_pair_num), kvol, cap,
                                                       None if defer else L.ptr(dw), L.ptr(ws), ws.numel(), L.stream()),
                    'sparse_conv_wgrad')
            return None if defer else out.to(filters.dtype)

        if _overlap is not None:
            # the weight gradient feeds nothing until the optimizer: run it on a side stream next to the
            # LN-backward / dgrad chain of the earlier layers (both are latency bound at one wave of work)
            side, cur = _overlap.side, torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                filters_bp = run()
            for t in (x, dy, pairs, indice_pair_num):
                t.record_stream(side)
            _overlap.used = True
        else:
            filters_bp = run()
    if need_input_grad:
        rb, (table, mask, rows) = _tables_for(indice_pairs, indice_pair_num, inverse, 'bwd', n_in,
                                              subm)
        nc = (cin + 15) // 16 * 16
        mode = 1 if (rb.subm and subm) else 2
        if mode == 1 and _fragment_major(rb, kd_out, nc):
            mode = 5  # fragment-major order for the tile kernel
        wn = _prep_weights(filters, mode, kd_out, nc)
        out_dtype = torch.bfloat16 if features.dtype == torch.bfloat16 else torch.float32
        if (_ln_link is not None and FUSE_LN_BACKWARD and mode == 5 and nc == cin and nc in (32, 64)
                and out_dtype == torch.bfloat16 and _ln_link.conv_out is not None
                and tuple(_ln_link.conv_out.shape) == (rows, nc)):
            # the block in front (conv -> LN -> act) gets its LayerNorm backward in this kernel's epilogue: what
            # leaves is the gradient of ITS conv output (functional.LnBackwardLink)
            prows = int(L.lib.ococc_sparse_conv_tile_lnbwd_partial_rows(rows, kd_out, nc))
            partials = L.empty((prows, 2 * nc), torch.float32, features.device)
            gin = L.empty((rows, nc), torch.bfloat16, features.device)
            kvol = wn.shape[0]
            _launch('tile_lnbwd', kd_out, nc, lambda: L.check(L.lib.ococc_sparse_conv_tile_lnbwd_bf16(
                L.ptr(dy), dy.size(0), kd_out, L.ptr(wn), kvol, nc, L.ptr(table), kvol // 2, rows,
                L.ptr(_ln_link.conv_out), L.ptr(_ln_link.stats), L.ptr(_ln_link.g32), L.ptr(_ln_link.b32),
                int(_ln_link.act), L.ptr(gin), L.ptr(partials), prows, L.stream()), 'sparse_conv_tile_lnbwd'))
            _ln_link.fused, _ln_link.partials, _ln_link.rows = True, partials, prows
            _ln_link.expect = (gin.data_ptr(), gin._version)   # (the engine may add a second consumer's gradient in place)
            return gin, filters_bp
        if (_ln_link is not None and FUSE_LN_BACKWARD and mode == 1 and nc == cin and nc in (32, 64)
                and out_dtype == torch.bfloat16 and _ln_link.conv_out is not None
                and tuple(_ln_link.conv_out.shape) == (rows, nc) and _use_sorted_kernel(rb, kd_out, nc)
                and 0 < rows < ORDER_MAX_ROWS):
            # the same fusion on the neighbour-pattern-order kernel (the finished f32 row sits in four lanes' accumulators)
            rec, hdr = row_order(rb, table, rows)
            prows = int(L.lib.ococc_sparse_conv_sorted_lnbwd_partial_rows(rows))
            partials = L.empty((prows, 2 * nc), torch.float32, features.device)
            gin = L.empty((rows, nc), torch.bfloat16, features.device)
            kvol = wn.shape[0]
            _launch('sorted_lnbwd', kd_out, nc, lambda: L.check(L.lib.ococc_sparse_conv_sorted_lnbwd_bf16(
                L.ptr(dy), dy.size(0), kd_out, L.ptr(wn), kvol, nc, L.ptr(table), L.ptr(rec), L.ptr(hdr), rows,
                L.ptr(_ln_link.conv_out), L.ptr(_ln_link.stats), L.ptr(_ln_link.g32), L.ptr(_ln_link.b32),
                int(_ln_link.act), L.ptr(gin), L.ptr(partials), prows, L.stream()), 'sparse_conv_sorted_lnbwd'))
            _ln_link.fused, _ln_link.partials, _ln_link.rows = True, partials, prows
            _ln_link.expect = (gin.data_ptr(), gin._version)
            return gin, filters_bp
        gin = _gather_gemm(dy, wn, table, mask, rows, None, out_dtype, rb if mode in (1, 5) else None)
        input_bp = gin if nc == cin else gin[:, :cin].contiguous()
    return input_bp, filters_bp
