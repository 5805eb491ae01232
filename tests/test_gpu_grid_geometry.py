"""ococc_object_grid_geometry_f32 (three launches, per-grid bitmaps in LDS) against the general path it replaces --
voxelize_scatter_mean(static=True) + get_indice_pairs(3x3x3 sub-manifold), both pinned to the reference's own C++
(tests/golden/voxelize.npz, tests/golden/rulebook.npz) -- and against the oracle: every output tensor bit for bit."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _points(B, per_grid, seed, half=4.0, ragged=False, empty_grid=None):
    g = torch.Generator().manual_seed(seed)
    counts = [per_grid] * B
    if ragged:
        counts = [int(v) for v in torch.randint(1, 2 * per_grid, (B,), generator=g)]
    if empty_grid is not None:
        counts[empty_grid] = 0
    n = sum(counts)
    xyz = (torch.rand(n, 3, generator=g) * 2 - 1) * half * 1.02       # a few points outside: clamped
    xyz[: n // 50] = xyz[n // 50: 2 * (n // 50)]                          # exact duplicates -> shared cells
    feats = torch.randn(n, 16, generator=g)
    bidx = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(counts))
    return xyz, feats, bidx


CASES = [dict(B=64, per_grid=2000, shape=(40, 40, 40), vs=0.2, half=4.0),              # the benchmark batch
         dict(B=5, per_grid=300, shape=(40, 40, 40), vs=0.2, half=4.0, ragged=True),
         dict(B=4, per_grid=900, shape=(40, 40, 40), vs=0.2, half=4.0, empty_grid=2),
         dict(B=3, per_grid=5000, shape=(16, 16, 16), vs=0.5, half=4.0),                # dense: many points per cell
         dict(B=2, per_grid=6000, shape=(64, 80, 80), vs=0.1, half=4.0),                # configs[4] cell size (z cut)
         dict(B=1, per_grid=1, shape=(40, 40, 40), vs=0.2, half=4.0),
         dict(B=70, per_grid=150, shape=(40, 40, 40), vs=0.2, half=4.0),               # > 256 count workgroups: the prefix launch
         dict(B=2, per_grid=30000, shape=(16, 16, 16), vs=0.5, half=4.0)]              # 52 k padding rows: several turns of the padding workgroups


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('slices', [1, 4, 8, 13, None])
def test_fused_geometry_equals_general_path(dev, case, slices):
    from objectcentricocccompletion_amd.spconv import ops
    from objectcentricocccompletion_amd.voxel import object_grid_geometry, voxelize_scatter_mean
    B, shape, vs = case['B'], list(case['shape']), case['vs']
    xyz, feats, bidx = _points(B, case['per_grid'], seed=B * 7 + (slices or 0), half=case['half'], ragged=case.get('ragged', False),
                               empty_grid=case.get('empty_grid'))
    rng = [-case['half']] * 2 + [-shape[0] * vs / 2] + [case['half']] * 2 + [shape[0] * vs / 2]
    xyz[:, 2] = xyz[:, 2] * (shape[0] * vs / 2) / case['half']
    xyz, feats, bidx = xyz.to(dev), feats.to(dev), bidx.to(dev)
    vsz = [vs, vs, vs]
    ref = voxelize_scatter_mean(xyz, bidx, feats, vsz, rng, shape, B, static=True, out_dtype=torch.bfloat16)
    rf, rc, rinv, rcnt, rmeta = ref
    _, rpairs, rnum = ops.get_indice_pairs(rc, B, shape, 3, subm=True)
    got = object_grid_geometry(xyz, bidx, feats, vsz, rng, shape, B, out_dtype=torch.bfloat16, slices=slices)
    assert got is not None
    gf, gc, ginv, gcnt, gmeta, gpairs, gnum = got
    torch.cuda.synchronize()
    nv = int(rmeta[0])
    assert gmeta.tolist() == rmeta.tolist() and nv > 0
    assert torch.equal(gc, rc) and torch.equal(ginv, rinv) and torch.equal(gcnt, rcnt)
    assert torch.equal(gnum, rnum)
    rt, rmask, _ = rpairs._ococc.tables[(False, 'fwd')]
    gt, gmask, _ = gpairs._ococc.tables[(False, 'fwd')]
    assert torch.equal(gt, rt)                                    # offset-major neighbour table, -1 padding rows
    assert torch.equal(gmask, rmask)
    for k in range(27):
        c = int(rnum[k])
        assert torch.equal(gpairs[k, :, :c], rpairs[k, :, :c]), k     # CPU-functor order (tails unwritten in static form)
    # features: rows with one or two points exact, the others to the last bits (float atomics on both sides)
    single = rcnt <= 2
    assert torch.equal(gf[single], rf[single])
    assert torch.allclose(gf.float(), rf.float(), rtol=1e-2, atol=1e-6)
    # the grid tag serves a later rulebook on the same rows (another indice_key)
    _, p2, n2 = ops.get_indice_pairs(gc, B, shape, 3, subm=True)
    assert torch.equal(n2, rnum)
    # and against the oracle on the small cases
    if xyz.shape[0] <= 20000:
        zyx = O.dynamic_voxelize(xyz.cpu().numpy(), vsz, rng)
        coors = np.concatenate([bidx.cpu().numpy()[:, None], zyx], 1)
        ofe, oco, oinv, ocnt = O.dynamic_scatter(feats.cpu().numpy(), coors, 'mean')
        assert np.array_equal(gc[:nv].cpu().numpy(), oco) and np.array_equal(ginv.cpu().numpy(), oinv)
        ep, en = O.subm_rulebook(oco, B, shape)
        assert np.array_equal(gnum.cpu().numpy(), en)
        for k in range(27):
            assert np.array_equal(gpairs[k, :, :en[k]].cpu().numpy(), ep[k, :, :en[k]])


def test_unsorted_batch_index_is_reported(dev):
    from objectcentricocccompletion_amd.voxel import object_grid_geometry
    xyz, feats, bidx = _points(4, 500, seed=1)
    bidx = bidx.clone()
    bidx[100] = 3                                                   # a stray point inside grid 0's segment
    got = object_grid_geometry(xyz.to(dev), bidx.to(dev), feats.to(dev), [0.2] * 3, [-4] * 3 + [4] * 3, [40] * 3, 4)
    torch.cuda.synchronize()
    assert int(got[4][1]) == 1 and int(got[2][100]) == -1


def test_encoder_uses_the_fused_geometry_and_matches(dev):
    """SubMOccEncoder.geometry(static=True) takes the three-launch path; the step built on it equals the general one."""
    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
    torch.manual_seed(0)
    B, P = 6, 700
    xyz, feats, bidx = synthetic_object_grids(B, P, seed=5, device=dev)
    from objectcentricocccompletion_amd.spconv import ops
    model = SubMOccEncoder(grouped_points=True).to(dev)
    outs = []
    keep = ops.DEFAULT_PAIRS_PER_ROW
    try:
        ops.DEFAULT_PAIRS_PER_ROW = 1.6   # (one kernel family for both passes: the device-side estimate lands in between)
        for grouped in (True, False):
            model.grouped_points = grouped
            model.zero_grad(set_to_none=True)
            geo = model.geometry(xyz, feats, bidx, B, static=True)
            assert hasattr(geo, 'meta') == grouped
            out = model(geometry=geo)
            out.features.float().pow(2).mean().backward()
            torch.cuda.synchronize()
            outs.append((out.features.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    finally:
        ops.DEFAULT_PAIRS_PER_ROW = keep
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('case', [CASES[0], CASES[1], CASES[2], CASES[3], CASES[4], CASES[6], CASES[7]])
@pytest.mark.parametrize('slices', [4, 13, None])
def test_emit_leaves_the_row_order_records(dev, case, slices):
    """with a sparse density on record the emit kernel also writes the neighbour-pattern order's row records: the order
    built from them is a valid one (every row once, masks and the two lowest entries as in the table, classes 3+ / 2 /
    1 / 0 neighbours in that sequence with the same sizes as the order built from the table alone).  Two ways to the
    slots: the placing pass launched by the same call, or by the caller from the row records (ORDER_SIDE_STREAM)."""
    from objectcentricocccompletion_amd.spconv import ops
    from objectcentricocccompletion_amd.voxel import object_grid_geometry
    B, shape, vs = case['B'], list(case['shape']), case['vs']
    xyz, feats, bidx = _points(B, case['per_grid'], seed=B * 5 + (slices or 0), half=case['half'], ragged=case.get('ragged', False),
                               empty_grid=case.get('empty_grid'))
    rng = [-case['half']] * 2 + [-shape[0] * vs / 2] + [case['half']] * 2 + [shape[0] * vs / 2]
    xyz[:, 2] = xyz[:, 2] * (shape[0] * vs / 2) / case['half']
    xyz, feats, bidx = xyz.to(dev), feats.to(dev), bidx.to(dev)
    for side in (False, True):
        keep = ops.DEFAULT_PAIRS_PER_ROW, ops.ORDER_SIDE_STREAM
        try:
            ops.DEFAULT_PAIRS_PER_ROW, ops.ORDER_SIDE_STREAM = 1.8, side
            got = object_grid_geometry(xyz, bidx, feats, [vs] * 3, rng, shape, B, out_dtype=torch.bfloat16, slices=slices)
        finally:
            ops.DEFAULT_PAIRS_PER_ROW, ops.ORDER_SIDE_STREAM = keep
        pairs = got[5]
        rb = pairs._ococc
        table, _, rows = rb.tables[(False, 'fwd')]
        assert len(rb.orders) == 1
        rec, hdr = ops.row_order(rb, table, rows)
        _check_order(rec, hdr, table, rows)


def _check_order(rec, hdr, table, rows):
    from objectcentricocccompletion_amd.spconv import ops
    rec, hdr = rec.cpu().long(), hdr.cpu().tolist()
    tab = table.cpu().long()
    perm, smask = rec[:, 0], rec[:, 1] & 0xffffffff
    assert torch.equal(perm.sort().values, torch.arange(rows))
    tmask = torch.zeros(rows, dtype=torch.long)
    for k in range(27):
        tmask |= (tab[k] >= 0).long() << k
    assert torch.equal(smask, tmask[perm])
    nb = smask & ~(1 << 13)
    low1 = nb & -nb
    nb2 = nb & (nb - 1)
    low2 = nb2 & -nb2
    for col, low in ((2, low1), (3, low2)):
        want = torch.full((rows,), -1, dtype=torch.long)
        has = low != 0
        kk = torch.tensor([int(v).bit_length() - 1 for v in low.tolist()])
        want[has] = tab[kk[has], perm[has]]
        assert torch.equal(rec[:, col], want)
    pc = torch.tensor([bin(int(v)).count('1') for v in nb.tolist()])
    cls = 3 - pc.clamp(max=3)
    assert bool((cls[1:] >= cls[:-1]).all())
    # the same header as the order built by the stand-alone counting pass
    rb2 = ops.RulebookTables(True, 27)
    _, hdr2 = ops.row_order(rb2, table, rows)
    assert hdr == hdr2.cpu().tolist()
