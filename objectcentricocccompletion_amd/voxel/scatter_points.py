"""Point -> voxel / RoI scatter-reduce -- host mirror of
mmdet3d/ops/voxel/scatter_points.py:9-107 (dynamic_scatter / DynamicScatter) and of
the torch_scatter calls inside scatter_v2 (mmdet3d/ops/sst/sst_ops.py:171-174).

Kernels: ococc_grid_unique_i32 (sorted unique rows + inverse + counts),
ococc_segment_reduce_f32 / _bwd_f32.
"""
import os

import torch
from torch import nn
from torch.autograd import Function

from .. import _lib as L


def grid_unique(coors, dims=None, static=False):
    """Sorted unique rows of non-negative int coordinates.

    Returns (out_coors [U,ndim] int32, inv [N] int32 (-1 for dropped rows), counts [U] int32).
    ``dims`` (exclusive upper bound per column) avoids a device read-back; without it the
    bounds come from coors.amax (one sync, as torch.unique has anyway).

    ``static=True`` is the fixed-capacity form for HIP-graph capture: nothing is read back, the
    outputs keep their full capacity U = min(N, prod(dims)), rows past the device-side count are
    -1 in out_coors / 0 in counts, and a fourth value ``meta`` (int32 [2] on the device:
    [num_unique, out_of_bounds flag]) is returned for the caller to inspect when it likes."""
    L.require_device(coors)
    squeeze = coors.dim() == 1
    if coors.size(0) == 0:
        z = coors.new_zeros((0,), dtype=torch.int32)
        return coors.to(torch.int32), z, z.clone()
    c = coors.reshape(coors.size(0), -1)
    if c.dtype != torch.int32:
        c = c.to(torch.int32)
    c = c.contiguous()
    n, ndim = c.shape
    if dims is None:
        if static:
            raise L.OcoccError('grid_unique(static=True) needs dims (no device read-back)')
        dims = [int(v) + 1 for v in c.amax(0).clamp_min(0).tolist()]
    dims = [int(d) for d in dims]
    assert len(dims) == ndim
    cells = 1
    for d in dims:
        cells *= d
    cap = min(n, cells)
    nbytes = L.lib.ococc_grid_unique_workspace_bytes(ndim, L.i4(dims))
    if nbytes < 0:
        raise L.OcoccError(f'grid_unique: coordinate space {dims} too large for the bitmap plan')
    ws = L.workspace(nbytes, c.device)
    if static:
        out_coors = torch.full((cap, ndim), -1, dtype=torch.int32, device=c.device)
    else:
        out_coors = torch.empty((cap, ndim), dtype=torch.int32, device=c.device)
    inv = torch.empty((n,), dtype=torch.int32, device=c.device)
    counts = torch.empty((cap,), dtype=torch.int32, device=c.device)
    meta = torch.empty(2, dtype=torch.int32, device=c.device)  # [num_unique, status], zeroed by the C side
    L.check(L.lib.ococc_grid_unique_i32(L.ptr(c), n, ndim, L.i4(dims), L.ptr(out_coors), cap,
                                        L.ptr(inv), L.ptr(counts), meta.data_ptr(),
                                        meta.data_ptr() + 4, L.ptr(ws), ws.numel(), L.stream()),
            'grid_unique')
    def tag(t):
        # the workspace still holds the cell bitmap + prefix of exactly these rows: a sub-manifold
        # rulebook over the same grid can reuse them (spconv.ops.get_indice_pairs)
        if ndim == 4:
            bo, po = L.c_i64(), L.c_i64()
            L.check(L.lib.ococc_grid_unique_workspace_layout(ndim, L.i4(dims), bo, po), 'grid_unique_layout')
            t._ococc_grid = (ws, int(bo.value), int(po.value), tuple(dims), bool(static))
        return t
    if static:
        inv._ococc_counts = counts
        return (out_coors[:, 0] if squeeze else tag(out_coors)), inv, counts, meta
    num, status = meta.tolist()
    if status:
        raise L.OcoccError(f'grid_unique: a coordinate is outside the declared bounds {dims}')
    out_coors = out_coors[:num]
    if squeeze:
        out_coors = out_coors[:, 0]
    else:
        tag(out_coors)
    counts = counts[:num]
    inv._ococc_counts = counts  # rides along so later reductions skip the counting pass
    return out_coors, inv, counts


class _SegmentReduce(Function):
    """feats [N,C] f32, inv [N] int32 -> [G,C]; max routes the gradient to the smallest
    row index attaining the maximum (scatter_points_cuda.cu:136-179)."""

    @staticmethod
    def forward(ctx, feats, inv, num_segments, mode, counts):
        L.require_device(feats, inv)
        code = L.REDUCE[mode]
        assert feats.dtype == torch.float32, 'segment reduce computes in float32'
        feats = feats.contiguous()
        n, c = feats.shape
        dev = feats.device
        if counts is None and (code == 1 or code == 2):
            counts = torch.empty((num_segments,), dtype=torch.int32, device=dev)
            L.check(L.lib.ococc_segment_count_i32(L.ptr(inv), n, L.ptr(counts), num_segments,
                                                  L.stream()), 'segment_count')
        out = torch.empty((num_segments, c), dtype=torch.float32, device=dev)
        need_arg = code == 2 and ctx.needs_input_grad[0]
        arg = torch.empty((num_segments, c), dtype=torch.int32, device=dev) if need_arg else None
        L.check(L.lib.ococc_segment_reduce_f32(L.ptr(feats), L.ptr(inv), n, c, code,
                                               L.ptr(counts), L.ptr(out), L.ptr(arg),
                                               num_segments, L.stream()), 'segment_reduce')
        ctx.code, ctx.shape, ctx.num_segments = code, (n, c), num_segments
        ctx.save_for_backward(inv, counts, arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        inv, counts, arg = ctx.saved_tensors
        n, c = ctx.shape
        grad_out = grad_out.contiguous().float()
        grad = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
        L.check(L.lib.ococc_segment_reduce_bwd_f32(L.ptr(grad_out), L.ptr(inv), n, c, ctx.code,
                                                   L.ptr(counts), L.ptr(arg), L.ptr(grad),
                                                   ctx.num_segments, L.stream()),
                'segment_reduce_bwd')
        return grad, None, None, None, None


def segment_reduce(feats, inv, num_segments, mode, counts=None):
    """Reduce rows of feats into num_segments rows following the dense inverse map inv."""
    if counts is None:
        counts = getattr(inv, '_ococc_counts', None)  # cached by grid_unique / unique_with_inverse
    if inv.dtype != torch.int32:
        inv = inv.to(torch.int32)
    return _SegmentReduce.apply(feats, inv.contiguous(), int(num_segments), mode, counts)


class _GatherRows(Function):
    """rows[inv]: the "map voxel features back to points" step of SIRLayer
    (voxel_encoder.py:758-760,811).  Backward is a segment SUM of the incoming rows -- the HIP
    run-length reduction instead of torch's sort-based index backward."""

    @staticmethod
    def forward(ctx, rows, inv):
        ctx.save_for_backward(inv)
        ctx.num_rows = rows.size(0)
        return rows.index_select(0, inv.long() if inv.dtype != torch.int64 else inv)

    @staticmethod
    def backward(ctx, grad):
        (inv,) = ctx.saved_tensors
        grad = grad.contiguous().float()
        n, c = grad.shape
        out = torch.empty((ctx.num_rows, c), dtype=torch.float32, device=grad.device)
        i32 = inv if inv.dtype == torch.int32 else inv.to(torch.int32)
        L.check(L.lib.ococc_segment_reduce_f32(L.ptr(grad), L.ptr(i32), n, c, 0, None, L.ptr(out), None,
                                               ctx.num_rows, L.stream()), 'gather_rows_bwd')
        return out, None


def gather_rows(rows, inv):
    return _GatherRows.apply(rows, inv)


def dynamic_scatter(feats, coors, reduce_type='max', grid_shape=None, static=False):
    """mmdet3d/ops/voxel/scatter_points.py:9-50: (voxel_feats [M,C], voxel_coors [M,ndim]).
    Rows of coors with a negative entry are dropped; output rows are in sorted order.
    ``static=True``: fixed-capacity outputs (M = number of points; unused rows have -1 coordinates
    and zero features), no device read-back -- see grid_unique."""
    if static:
        voxel_coors, inv, counts, _ = grid_unique(coors, grid_shape, static=True)
    else:
        voxel_coors, inv, counts = grid_unique(coors, grid_shape)
    voxel_feats = segment_reduce(feats, inv, voxel_coors.size(0), reduce_type, counts)
    return voxel_feats, voxel_coors.to(coors.dtype)


class _MeanOfPointsGrad(Function):
    """Ties the voxel means computed by ococc_voxelize_scatter_mean_f32 to the point features they
    came from: d feats[i] = d voxel[inv[i]] / counts[inv[i]] (the MEAN branch of _SegmentReduce)."""

    @staticmethod
    def forward(ctx, feats, vfeats, inv, counts):
        ctx.save_for_backward(inv, counts)
        ctx.shape = tuple(feats.shape)
        return vfeats.view_as(vfeats)

    @staticmethod
    def backward(ctx, grad_out):
        inv, counts = ctx.saved_tensors
        n, c = ctx.shape
        grad_out = grad_out.contiguous().float()
        grad = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
        L.check(L.lib.ococc_segment_reduce_bwd_f32(L.ptr(grad_out), L.ptr(inv), n, c, L.REDUCE['mean'],
                                                   L.ptr(counts), None, L.ptr(grad), counts.size(0), L.stream()),
                'voxelize_scatter_mean_bwd')
        return grad, None, None, None


def voxelize_scatter_mean(points, batch_idx, feats, voxel_size, coors_range, grid_zyx, batch_size,
                          static=False, out_dtype=torch.float32):
    """voxelization(points, voxel_size, coors_range, -1, -1) -> coors = cat(batch_idx, zyx) ->
    dynamic_scatter(feats, coors, 'mean') in ONE C-ABI call (ococc_voxelize_scatter_mean_f32): same
    results, same row order, no [N,3] / [N,4] coordinate tensors, no zero fills, float atomics only
    for the points that share a cell with an earlier one.

    Returns (voxel_feats [M,C] in out_dtype (float32 or bfloat16), voxel_coors [M,4] int32 (b,z,y,x),
    inv [N] int32, counts [M] int32, meta int32[2] on the device = [num_voxels, status]).
    ``static=True``: M is the fixed capacity min(N, cells) and nothing is read back (rows past
    meta[0] have -1 coordinates, count 0, zero features); otherwise M = num_voxels (one sync).
    The returned coordinates carry the grid bitmap tag spconv.ops.get_indice_pairs reuses."""
    L.require_device(points, batch_idx, feats)
    want16 = out_dtype == torch.bfloat16
    assert want16 or out_dtype == torch.float32
    assert points.dtype == torch.float32 and feats.dtype == torch.float32
    pts, fts = points.detach().contiguous(), feats.detach().contiguous()
    bidx = batch_idx if batch_idx.dtype == torch.int32 else batch_idx.to(torch.int32)
    bidx = bidx.contiguous()
    n, c = fts.shape
    dev = fts.device
    grid_zyx = [int(v) for v in grid_zyx]
    dims = [int(batch_size)] + grid_zyx
    cap = min(n, dims[0] * dims[1] * dims[2] * dims[3])
    nbytes = L.lib.ococc_voxelize_scatter_workspace_bytes(n, int(batch_size), L.i3(grid_zyx))
    if nbytes < 0:
        raise L.OcoccError(f'voxelize_scatter_mean: grid {dims} too large for the bitmap plan')
    ws = L.workspace(nbytes, dev)
    coors = L.empty((cap, 4), torch.int32, dev)
    inv = L.empty((n,), torch.int32, dev)
    counts = L.empty((cap,), torch.int32, dev)
    out = L.empty((cap, c), torch.float32, dev)
    out16 = L.empty((cap, c), torch.bfloat16, dev) if want16 else None
    meta = L.empty((2,), torch.int32, dev)
    L.check(L.lib.ococc_voxelize_scatter_mean_f32(
        L.ptr(pts), pts.size(1), L.ptr(bidx), n, L.ptr(fts), c, L.f3(voxel_size), L.f6(coors_range),
        int(batch_size), L.i3(grid_zyx), L.ptr(coors), cap, L.ptr(inv), L.ptr(counts), L.ptr(out), L.ptr(out16),
        meta.data_ptr(), meta.data_ptr() + 4, L.ptr(ws), ws.numel(), L.stream()), 'voxelize_scatter_mean')
    vfeats = out16 if want16 else out
    if not static:
        num, status = meta.tolist()
        if status:
            raise L.OcoccError(f'voxelize_scatter_mean: a batch index is outside [0, {batch_size})')
        vfeats, coors, counts = vfeats[:num], coors[:num], counts[:num]
    if feats.requires_grad and torch.is_grad_enabled():
        vfeats = _MeanOfPointsGrad.apply(feats, vfeats, inv, counts)
    bo, po = L.c_i64(), L.c_i64()
    L.check(L.lib.ococc_grid_unique_workspace_layout(4, L.i4(dims), bo, po), 'grid_unique_layout')
    coors._ococc_grid = (ws, int(bo.value), int(po.value), tuple(dims), bool(static))
    inv._ococc_counts = counts
    return vfeats, coors, inv, counts, meta


# Row slices (workgroups of the emit kernel) per grid: 0 = from the points per grid, so that a slice holds fewer rows
# than the emit kernel's 256 threads (a slice with more takes a second round of the whole row loop: at 8 slices for
# 2000 points a quarter of the workgroups did, and the kernel waited for them: 34.4 vs 28.8 us)
GEOMETRY_SLICES = int(os.environ.get('OCOCC_GEO_SLICES', 0))


def object_grid_geometry(points, batch_idx, feats, voxel_size, coors_range, grid_zyx, batch_size,
                         out_dtype=torch.float32, slices=None):
    """voxelize_scatter_mean(static=True) followed by spconv.ops.get_indice_pairs(3x3x3 sub-manifold) on its rows, in
    three launches (ococc_object_grid_geometry_f32) instead of ten, for points that arrive GROUPED BY GRID (batch_idx
    non-decreasing).  Same tensors, same values: returns (voxel_feats, voxel_coors, inv, counts, meta, indice_pairs,
    indice_pair_num); the pair tensor carries the gather tables the convolutions read, the coordinates the grid tag,
    exactly as the two separate calls leave them.  None when the shape is outside the kernel's plan (cells per grid
    not a multiple of 32 or above 512 Ki): the caller then takes the general path."""
    from ..spconv import ops as sp_ops
    L.require_device(points, batch_idx, feats)
    want16 = out_dtype == torch.bfloat16
    assert want16 or out_dtype == torch.float32
    assert points.dtype == torch.float32 and feats.dtype == torch.float32
    pts, fts = points.detach().contiguous(), feats.detach().contiguous()
    bidx = (batch_idx if batch_idx.dtype == torch.int32 else batch_idx.to(torch.int32)).contiguous()
    n, c = fts.shape
    dev = fts.device
    grid_zyx = [int(v) for v in grid_zyx]
    dims = [int(batch_size)] + grid_zyx
    cap = min(n, dims[0] * dims[1] * dims[2] * dims[3])
    if slices is None:
        slices = GEOMETRY_SLICES or min(16, max(1, -(-points.size(0) // max(int(batch_size), 1) // 176)))
    nbytes = L.lib.ococc_object_grid_geometry_workspace_bytes(n, int(batch_size), L.i3(grid_zyx), int(slices))
    if nbytes < 0 or n == 0:
        return None
    ws = L.workspace(nbytes, dev)
    coors = L.empty((cap, 4), torch.int32, dev)
    inv = L.empty((n,), torch.int32, dev)
    counts = L.empty((cap,), torch.int32, dev)
    out = L.empty((cap, c), torch.float32, dev)
    out16 = L.empty((cap, c), torch.bfloat16, dev) if want16 else None
    meta = L.empty((2,), torch.int32, dev)
    nbr_t = L.empty((27, cap), torch.int32, dev)
    mask = L.empty(((cap + 15) // 16,), torch.int32, dev)
    pairs = L.empty((27, 2, cap), torch.int32, dev)
    num = L.empty((27,), torch.int32, dev)
    # (sparse rulebooks: the emit kernel also builds the neighbour-pattern row order, spconv.ops.row_order -- the finished
    # order unless its placing pass is wanted on a side stream, then the row records only)
    rowrec = L.empty((cap, 4), torch.int32, dev) if sp_ops.new_rulebook_wants_order(cap) else None
    order = None
    if rowrec is not None and not sp_ops.ORDER_SIDE_STREAM:
        order = (L.empty((cap, 4), torch.int32, dev), L.empty((8,), torch.int32, dev))
    if rowrec is not None:
        rowrec._ococc_counters = sp_ops.order_counters(dev)   # (this stream's: whoever places the records uses the same)
    L.check(L.lib.ococc_object_grid_geometry_order_f32(
        L.ptr(pts), pts.size(1), L.ptr(bidx), n, L.ptr(fts), c, L.f3(voxel_size), L.f6(coors_range), int(batch_size),
        L.i3(grid_zyx), int(slices), L.ptr(coors), cap, L.ptr(inv), L.ptr(counts), L.ptr(out), L.ptr(out16),
        meta.data_ptr(), meta.data_ptr() + 4, L.ptr(nbr_t), L.ptr(mask), L.ptr(pairs), L.ptr(num), L.ptr(ws), ws.numel(),
        L.ptr(rowrec._ococc_counters) if rowrec is not None else None, L.ptr(rowrec),
        L.ptr(order[0]) if order else None, L.ptr(order[1]) if order else None, sp_ops.SORTED_TILES[0], sp_ops.SORTED_TILES[1],
        L.stream()), 'object_grid_geometry')
    vfeats = out16 if want16 else out
    bo, po = L.c_i64(), L.c_i64()
    L.check(L.lib.ococc_grid_unique_workspace_layout(4, L.i4(dims), bo, po), 'grid_unique_layout')
    coors._ococc_grid = (ws, int(bo.value), int(po.value), tuple(dims), True)
    inv._ococc_counts = counts
    sp_ops.attach_subm_tables(pairs, nbr_t, mask, cap, 27, num=num, rowrec=rowrec, order=order)   # (num: observed by the density tracker)
    pairs._ococc_keepalive = ws
    return vfeats, coors, inv, counts, meta, pairs, num


class DynamicScatter(nn.Module):
    """Same constructor / forward as mmdet3d/ops/voxel/scatter_points.py:53-107.  The
    reference loops over the batch in Python (:86-100); here the batch column is part of
    the key, which yields the same rows in the same order with one launch sequence."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points

    def forward_single(self, points, coors):
        reduce = 'mean' if self.average_points else 'max'
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce)

    def forward(self, points, coors):
        return self.forward_single(points, coors)

    def __repr__(self):
        return (f'{self.__class__.__name__}(voxel_size={self.voxel_size}, '
                f'point_cloud_range={self.point_cloud_range}, '
                f'average_points={self.average_points})')
