"""Host mirror of mmdet3d/ops/sst/sst_ops.py: scatter_v2 (:150-181), build_mlp (:333-360),
get_activation / get_activation_layer (:362-391).  The torch.unique + torch_scatter pair of
the reference becomes ococc_grid_unique_i32 + ococc_segment_reduce_f32 (HIP)."""
import torch
import torch.nn as nn

from ..norm import LayerNorm
from ..registry import build_norm_layer
from ..voxel.scatter_points import grid_unique, segment_reduce


def unique_with_inverse(coors, dims=None, return_counts=False):
    """torch.unique(coors, return_inverse=True, dim=0) for integer keys >= -1 (the reference
    feeds RoI / voxel indices where -1 marks "no group" and is kept as a group of its own,
    sorted first -- ococc_bbox_head.py:255-260).  Returns int32 tensors."""
    # (the two point encoders of OccBBoxHead group the same pooled points by the same RoI index: the second call finds
    # the first one's result on the key tensor -- one grouping pass and one read-back of the group count less per step)
    key = None if dims is None else tuple(int(d) for d in dims)
    memo = getattr(coors, '_ococc_unique', None)
    if memo is not None and memo[0] == key and memo[1] == coors._version:
        new_coors, inv, counts = memo[2]
    else:
        shifted = coors + 1  # -1 -> 0: the bitmap ranks only non-negative keys
        known = getattr(coors, '_ococc_num_groups', None)   # (point_pool: the producer already read the group count back)
        if known is not None and dims is not None:
            new_coors, inv, counts, _ = grid_unique(shifted, [int(d) + 1 for d in dims], static=True)
            new_coors, counts = new_coors[:known], counts[:known]
        else:
            new_coors, inv, counts = grid_unique(shifted, None if dims is None else [int(d) + 1 for d in dims])
        new_coors = new_coors - 1
        inv._ococc_counts = counts
        if not coors.requires_grad:
            coors._ococc_unique = (key, coors._version, (new_coors, inv, counts))
    if return_counts:
        return new_coors, inv, counts
    return new_coors, inv


def scatter_v2(feat, coors, mode, return_inv=True, min_points=0, unq_inv=None, new_coors=None,
               dims=None):
    """sst_ops.py:150-181.  ``dims`` (exclusive upper bound of each key column) is an
    extension that spares the device read-back torch.unique implies."""
    assert feat.size(0) == coors.size(0)
    if mode == 'avg':
        mode = 'mean'
    counts = None
    if unq_inv is None:
        new_coors, unq_inv, counts = unique_with_inverse(coors, dims, return_counts=True)
    else:
        assert new_coors is not None, 'please pass new_coors for interface consistency'
    if min_points > 0:
        if counts is None:
            counts = torch.bincount(unq_inv.long(), minlength=new_coors.size(0))
        valid_mask = counts[unq_inv.long()] >= min_points
        feat = feat[valid_mask]
        coors = coors[valid_mask]
        new_coors, unq_inv, counts = unique_with_inverse(coors, dims, return_counts=True)
    if mode not in ('max', 'mean', 'sum'):
        raise NotImplementedError
    new_feat = segment_reduce(feat.float(), unq_inv, new_coors.size(0), mode, counts)
    if not return_inv:
        return new_feat, new_coors
    return new_feat, new_coors, unq_inv


def get_activation(activation):
    """sst_ops.py:362-370."""
    if activation == 'relu':
        return torch.nn.functional.relu
    if activation == 'gelu':
        return torch.nn.functional.gelu
    if activation == 'glu':
        return torch.nn.functional.glu
    raise RuntimeError(F'activation should be relu/gelu, not {activation}.')


def get_activation_layer(act, dim=None):
    """sst_ops.py:372-391."""
    act = act.lower()
    if act == 'relu':
        return nn.ReLU(inplace=True)
    if act == 'gelu':
        return nn.GELU()
    if act == 'leakyrelu':
        return nn.LeakyReLU(inplace=True)
    if act == 'prelu':
        return nn.PReLU(num_parameters=dim)
    if act in ('swish', 'silu'):
        return nn.SiLU(inplace=True)
    if act == 'glu':
        return nn.GLU()
    if act == 'elu':
        return nn.ELU(inplace=True)
    raise NotImplementedError


def fuse_norm_act(norm_layer, act_layer):
    """LN followed by exact GELU -> one HIP pass; the activation slot keeps a parameter-free
    Identity so that child indices (state-dict keys) stay those of the reference."""
    if isinstance(norm_layer, LayerNorm) and isinstance(act_layer, nn.GELU):
        norm_layer.fused_act = 'gelu'
        return norm_layer, nn.Identity()
    return norm_layer, act_layer


from ..linear import Linear  # noqa: E402 (nn.Linear with a sliced weight gradient for 1e5-row inputs)


def build_mlp(in_channel, hidden_dims, norm_cfg, is_head=False, act='relu', bias=False, dropout=0):
    """sst_ops.py:333-360: Sequential of Sequential(Linear, norm, act[, Dropout]) blocks; with
    is_head the last entry is a bare Linear(bias=True).  Keys: <i>.0.weight, <i>.1.{weight,bias}."""
    layer_list = []
    last_channel = in_channel
    if isinstance(hidden_dims, int):
        hidden_dims = [hidden_dims, ]
    for i, c in enumerate(hidden_dims):
        act_layer = get_activation_layer(act, c)
        norm_layer = build_norm_layer(norm_cfg, c)[1]
        if i == len(hidden_dims) - 1 and is_head:
            layer_list.append(Linear(last_channel, c, bias=True))
        else:
            norm_layer, act_layer = fuse_norm_act(norm_layer, act_layer)
            sq = [Linear(last_channel, c, bias=bias), norm_layer, act_layer]
            if dropout > 0:
                from ..norm import FoldedDropout, LayerNorm
                if isinstance(norm_layer, LayerNorm) and isinstance(act_layer, nn.Identity):
                    # norm and activation are one kernel already; the dropout behind them joins it (no byte mask)
                    norm_layer.fused_dropout = float(dropout)
                    sq.append(FoldedDropout(dropout))
                else:
                    sq.append(nn.Dropout(dropout))
            layer_list.append(nn.Sequential(*sq))
        last_channel = c
    return nn.Sequential(*layer_list)


# ------------------------------------------------------------------------------------------
# SST window bookkeeping (sst_ops.py:26-148, 243-330)
# ------------------------------------------------------------------------------------------
def group_rank(keys, key_bound=None):
    """(conti, inner, counts): rank of each key among the distinct keys, stable rank of each element
    inside its group, and the group sizes.  HIP: ococc_group_rank_i32."""
    from .. import _lib as L
    L.require_device(keys)
    k32 = keys.to(torch.int32).contiguous()
    n = k32.numel()
    dev = keys.device
    if n == 0:
        z = torch.zeros((0,), dtype=torch.int32, device=dev)
        return z, z.clone(), z.clone()
    if key_bound is None:
        key_bound = int(k32.max().item()) + 1
    key_bound = max(int(key_bound), 1)
    ws = L.workspace(L.lib.ococc_group_rank_workspace_bytes(n, key_bound), dev)
    conti = torch.empty((n,), dtype=torch.int32, device=dev)
    inner = torch.empty((n,), dtype=torch.int32, device=dev)
    counts = torch.empty((n,), dtype=torch.int32, device=dev)
    meta = torch.zeros((2,), dtype=torch.int32, device=dev)
    L.check(L.lib.ococc_group_rank_i32(L.ptr(k32), n, key_bound, L.ptr(conti), L.ptr(inner), L.ptr(counts),
                                       meta.data_ptr(), meta.data_ptr() + 4, L.ptr(ws), ws.numel(), L.stream()),
            'group_rank')
    num, status = meta.tolist()
    if status:
        raise L.OcoccError(f'group_rank: a key is >= the declared bound {key_bound}')
    return conti, inner, counts[:num]


@torch.no_grad()
def get_inner_win_inds(win_inds):
    """Index of every voxel inside its window, 0..m-1 (sst_ops.py:243-263; the reference accepts any
    order within a window, ours is the stable one)."""
    return group_rank(win_inds)[1].to(win_inds.dtype)


class IngroupIndicesFunction(torch.autograd.Function):
    """The reference's wrapper around TorchEx's ingroup_indices (sst_ops.py:245-263): forward only, the result marked
    non-differentiable.  ``get_inner_win_inds`` above is the same computation."""

    @staticmethod
    def forward(ctx, group_inds):
        out_inds = group_rank(group_inds)[1].to(group_inds.dtype)
        ctx.mark_non_differentiable(out_inds)
        return out_inds

    @staticmethod
    def backward(ctx, g):
        return None


@torch.no_grad()
def get_inner_win_inds_deprecated(win_inds):
    """The torch-only formulation the reference keeps beside the TorchEx kernel (sst_ops.py:193-241): sort the window
    ids, number the members of every run, undo the sort.  Any order inside a window is a valid answer there (its sort is
    unstable); the sort here is stable, so the result equals get_inner_win_inds.  Host of the comparison in the tests,
    and usable on CPU tensors."""
    sort_inds, order = torch.sort(win_inds, stable=True)
    n = win_inds.numel()
    if n == 0:
        return torch.zeros_like(win_inds)
    pos = torch.arange(n, device=win_inds.device, dtype=torch.long)
    new_run = torch.ones(n, dtype=torch.bool, device=win_inds.device)
    new_run[1:] = sort_inds[1:] != sort_inds[:-1]
    run_start = torch.cummax(torch.where(new_run, pos, torch.zeros_like(pos)), 0)[0]
    inner = torch.empty(n, dtype=torch.long, device=win_inds.device)
    inner[order] = pos - run_start
    return inner.to(win_inds.dtype)


def filter_almost_empty(pts_coors, min_points=5):
    """Mask of the points whose voxel holds at least ``min_points`` points (sst_ops.py:183-190; the reference's body
    names an undefined ``coors`` in the counting branch -- the evident intent is restated here)."""
    if min_points > 0:
        _, unq_inv, unq_cnt = unique_with_inverse(pts_coors, return_counts=True)
        return unq_cnt.to(torch.long)[unq_inv.long()] >= min_points
    return torch.ones(len(pts_coors), device=pts_coors.device, dtype=torch.bool)


@torch.no_grad()
def make_continuous_inds(inds):
    """sst_ops.py:316-330: relabel window ids 0..num_windows-1 in sorted order."""
    return group_rank(inds)[0].to(inds.dtype)


@torch.no_grad()
def get_window_coors_both(coors, sparse_shape, window_shape):
    """get_window_coors for the unshifted and the shifted partition in ONE launch (ococc_sst_window_coors_i64):
    [(batch_win_inds, coors_in_win)] for shift 0 and 1.  coors [N,4] int64 (b,z,y,x) on the device."""
    from .. import _lib as L
    L.require_device(coors)
    import ctypes
    c = coors.contiguous()
    assert c.dtype == torch.int64 and c.dim() == 2 and c.size(1) == 4
    n = c.size(0)
    win = list(window_shape) if len(window_shape) == 3 else [window_shape[0], window_shape[1], sparse_shape[-1]]
    assert sparse_shape[2] < sparse_shape[0], 'Usually holds... in case of wrong order'
    ids = torch.empty((2, n), dtype=torch.int64, device=c.device)
    ciw = torch.empty((2, n, 3), dtype=torch.int64, device=c.device)
    i3 = lambda v: (ctypes.c_int32 * 3)(*[int(a) for a in v])
    L.check(L.lib.ococc_sst_window_coors_i64(L.ptr(c), n, i3(sparse_shape), i3(win), L.ptr(ids), L.ptr(ciw), L.stream()),
            'sst_window_coors')
    return [(ids[0], ciw[0]), (ids[1], ciw[1])]


@torch.no_grad()
def get_window_coors(coors, sparse_shape, window_shape, do_shift):
    """Window id of every voxel (unique in the batch) and its coordinate inside the window
    (sst_ops.py:266-313).  coors [N,4] = (b,z,y,x); integer elementwise math."""
    if coors.is_cuda and coors.dtype == torch.int64:
        return get_window_coors_both(coors, sparse_shape, window_shape)[1 if do_shift else 0]
    if len(window_shape) == 2:
        win_shape_x, win_shape_y = window_shape
        win_shape_z = sparse_shape[-1]
    else:
        win_shape_x, win_shape_y, win_shape_z = window_shape
    sparse_shape_x, sparse_shape_y, sparse_shape_z = sparse_shape
    assert sparse_shape_z < sparse_shape_x, 'Usually holds... in case of wrong order'
    import math
    max_x = int(math.ceil(sparse_shape_x / win_shape_x) + 1)
    max_y = int(math.ceil(sparse_shape_y / win_shape_y) + 1)
    max_z = int(math.ceil(sparse_shape_z / win_shape_z) + 1)
    if do_shift:
        shift_x, shift_y, shift_z = win_shape_x // 2, win_shape_y // 2, win_shape_z // 2
    else:
        shift_x, shift_y, shift_z = win_shape_x, win_shape_y, win_shape_z
    if sparse_shape_z == win_shape_z:
        shift_z = 0
    sx, sy, sz = coors[:, 3] + shift_x, coors[:, 2] + shift_y, coors[:, 1] + shift_z
    wx, wy, wz = sx // win_shape_x, sy // win_shape_y, sz // win_shape_z
    batch_win_inds = coors[:, 0] * (max_x * max_y * max_z) + wx * max_y * max_z + wy * max_z + wz
    coors_in_win = torch.stack([sz % win_shape_z, sy % win_shape_y, sx % win_shape_x], dim=-1)
    return batch_win_inds, coors_in_win


@torch.no_grad()
def get_flat2win_inds(batch_win_inds, voxel_drop_lvl, drop_info, debug=True, populations=None, key_bound=None):
    """Per drop level: slot of every voxel in the padded [num_windows * max_tokens] layout and the
    voxel positions of that level (sst_ops.py:26-63).  ``populations`` (a dict) receives per level the tokens of every
    window, which the group-rank kernel counts anyway: the key padding mask as a length.  ``key_bound``: an upper bound
    of the window ids when the caller knows one (saves the read-back of their maximum)."""
    out = {}
    levels = [int(dl) for dl in drop_info]
    if (key_bound is not None and batch_win_inds.is_cuda and batch_win_inds.numel() > 0 and min(levels) >= 0
            and (max(levels) + 1) * int(key_bound) < 2 ** 31):
        # ONE group-rank pass over the composite key (level, window) serves every level: its groups come out level by
        # level, so a level's windows are a contiguous range of ranks and its populations a slice of the counts; the
        # voxel lists of the levels are the pieces of one stable sort.  Two read-backs (group count; voxels and windows
        # per level) where the per-level loop below has two per level.
        nl = max(levels) + 1
        lv_raw = voxel_drop_lvl.long()
        lv = lv_raw.clamp(min=0)   # (a level of -1 -- a window population in no drop range -- is reported below, from the
        #                             same read-back; clamped here so that no rank is left unwritten in the meantime)
        conti, inner, counts = group_rank(lv * int(key_bound) + batch_win_inds.long(), nl * int(key_bound))
        # voxels by level: ONE stable sort of 8-bit keys (a level of -1 wraps to 255 and sorts last); the sorted keys
        # also give the voxels per level (binary searches for the level boundaries) -- an int64 sort plus an [n, levels]
        # comparison summed over n cost 0.2 ms per call
        if nl < 255:
            vals, order = torch.sort(lv_raw.to(torch.uint8), stable=True)
            bounds = torch.searchsorted(vals, torch.arange(nl + 1, device=lv.device, dtype=torch.uint8))
            per_level = bounds[1:] - bounds[:-1]
        else:
            order = torch.argsort(lv, stable=True)
            per_level = (lv_raw[:, None] == torch.arange(nl, device=lv.device)[None, :]).sum(0)
        ids = torch.arange(nl, device=lv.device)
        glvl = torch.zeros(counts.numel(), dtype=torch.long, device=lv.device).scatter_(0, conti.long(), lv)
        tally = torch.stack([per_level.long(), (glvl[:, None] == ids[None, :]).sum(0)]).tolist()
        if sum(tally[0]) != lv.numel():
            # the reference: assert (drop_lvl_per_voxel >= 0).all() (sst_input_layer_v2.py)
            raise ValueError(f'{lv.numel() - sum(tally[0])} voxels sit in windows whose population is in no drop range '
                             '(drop level -1): the drop_info ranges must cover every window size')
        vo = wo = 0
        for dl in range(nl):
            nv, nw = tally[0][dl], tally[1][dl]
            if dl in drop_info and nv > 0:
                idx = order[vo:vo + nv]
                max_tokens = drop_info[dl]['max_tokens']
                inner_l = inner[idx].long()
                if debug:
                    assert int(inner_l.max()) < max_tokens, f'Max inner inds({int(inner_l.max())}) larger(equal) than {max_tokens}'
                out[dl] = (((conti[idx].long() - wo) * max_tokens + inner_l), (idx,))
                if populations is not None:
                    populations[dl] = counts[wo:wo + nw].to(torch.int32)
            vo, wo = vo + nv, wo + nw
        return {dl: out[dl] for dl in drop_info if dl in out}   # (the reference's dict order)
    for dl in drop_info:
        where = torch.where(voxel_drop_lvl == dl)   # (the one read-back of this level: its voxel list)
        if where[0].numel() == 0:
            continue
        conti, inner, counts = group_rank(batch_win_inds[where[0]], key_bound)
        max_tokens = drop_info[dl]['max_tokens']
        if debug:
            assert int(inner.max()) < max_tokens, f'Max inner inds({int(inner.max())}) larger(equal) than {max_tokens}'
        out[dl] = ((conti.long() * max_tokens + inner.long()), where)
        if populations is not None:
            populations[dl] = counts.to(torch.int32)
    return out


def get_flat2win_inds_v2(batch_win_inds, voxel_drop_lvl, drop_info, debug=True, key_bound=None):
    pop = {}
    d = get_flat2win_inds(batch_win_inds, voxel_drop_lvl, drop_info, debug, populations=pop, key_bound=key_bound)
    d['voxel_drop_level'] = voxel_drop_lvl
    d['batching_info'] = drop_info
    d['_ococc_populations'] = pop
    return d


class LazyWindowDict(dict):
    """A per-drop-level dict of padded window tensors that is only built when somebody reads it.  The reference's input
    layer materialises the padded positional embedding and key padding masks up front (sst_input_layer_v2.py:122-126);
    the fused encoder layers read neither -- they take the embedding in flat token order and the window populations --
    so here the flat2window copies happen on first access."""

    def __init__(self, build):
        super().__init__()
        self._build = build

    def _fill(self):
        if self._build is not None:
            build, self._build = self._build, None
            super().update(build())

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __contains__(self, k):
        self._fill()
        return super().__contains__(k)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()

    def keys(self):
        self._fill()
        return super().keys()

    def values(self):
        self._fill()
        return super().values()

    def items(self):
        self._fill()
        return super().items()

    def get(self, k, default=None):
        self._fill()
        return super().get(k, default)


def flat2window(feat, voxel_drop_lvl, flat2win_inds_dict, drop_info, padding=0):
    """[N, C] -> per drop level [num_windows, max_tokens, C], padded (sst_ops.py:66-104)."""
    feat_dim = feat.shape[-1]
    out = {}
    for dl in drop_info:
        if dl not in flat2win_inds_dict:
            continue
        this_inds, flat_pos = flat2win_inds_dict[dl]
        max_tokens = drop_info[dl]['max_tokens']
        num_windows = int((this_inds // max_tokens).max().item()) + 1
        feat_3d = torch.full((num_windows * max_tokens, feat_dim), padding, dtype=feat.dtype, device=feat.device)
        feat_3d[this_inds] = feat[flat_pos[0]]
        out[dl] = feat_3d.reshape(num_windows, max_tokens, feat_dim)
    return out


def window2flat(feat_3d_dict, inds_dict):
    """Inverse of flat2window (sst_ops.py:106-131)."""
    first = feat_3d_dict[list(feat_3d_dict.keys())[0]]
    n = sum(inds_dict[dl][0].shape[0] for dl in inds_dict)
    out = torch.zeros((n, first.shape[-1]), device=first.device, dtype=first.dtype)
    for dl in feat_3d_dict:
        inds, flat_pos = inds_dict[dl]
        out[flat_pos[0]] = feat_3d_dict[dl].reshape(-1, first.shape[-1])[inds]
    return out


def window2flat_v2(feat_3d_dict, inds_dict):
    return window2flat(feat_3d_dict, {k: inds_dict[k] for k in inds_dict if not isinstance(k, str)})


def flat2window_v2(feat, inds_dict, padding=0):
    assert 'voxel_drop_level' in inds_dict, 'voxel_drop_level should be in inds_dict in v2 function'
    inds_v1 = {k: inds_dict[k] for k in inds_dict if not isinstance(k, str)}
    return flat2window(feat, inds_dict['voxel_drop_level'], inds_v1, inds_dict['batching_info'], padding=padding)
