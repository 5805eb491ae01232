"""Generate tests/golden/rulebook.npz from the REFERENCE's own CPU rulebook generators
(oracle/_ref/rulebook_ref.so = /root/reference/mmdet3d/ops/spconv/include/spconv/geometry.h
:144-297 + src/reordering.cc:21-50 behind oracle/rulebook_ref.cc, compiled by oracle/Makefile).
Runs only in the build container; the .npz (inputs + expected outputs) is committed, the
reference binary is not.

Stored per case: the input voxel indices [N,4] (b,z,y,x), the geometry, `num` [K], and the
pairs of every offset in the reference's order, compacted (`pairs_in`, `pairs_out`: the first
num[k] entries of indicePairs[k,0,:] / [k,1,:] for k = 0..K-1, concatenated -- the rest of the
reference's buffer is its -1 fill, spconv_ops.h:56-58), plus `out_indices` for the regular /
transposed cases.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
i64, ci = ctypes.c_int64, ctypes.c_int


def load_ref():
    lib = ctypes.CDLL(os.path.join(HERE, '_ref', 'rulebook_ref.so'))
    lib.ref_subm_rulebook.restype = i64
    lib.ref_conv_rulebook.restype = i64
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    return np.ascontiguousarray(a, np.int32)


def ref_subm(lib, idx, batch, shape, ksize=(3, 3, 3), dilation=(1, 1, 1)):
    idx = _i(idx)
    n, kvol = len(idx), int(np.prod(ksize))
    pairs = np.empty((kvol, 2, n), np.int32)
    num = np.empty((kvol,), np.int32)
    lib.ref_subm_rulebook(_p(idx), i64(n), ci(batch), _p(_i(shape)), _p(_i(ksize)), _p(_i(dilation)),
                          _p(pairs), _p(num))
    return pairs, num


def ref_conv(lib, idx, batch, out_shape, ksize, stride, padding, dilation, transpose):
    idx = _i(idx)
    n, kvol = len(idx), int(np.prod(ksize))
    pairs = np.empty((kvol, 2, n), np.int32)
    num = np.empty((kvol,), np.int32)
    outi = np.empty((max(n * kvol, 1), 4), np.int32)
    m = lib.ref_conv_rulebook(_p(idx), i64(n), ci(batch), _p(_i(out_shape)), _p(_i(ksize)), _p(_i(stride)),
                              _p(_i(padding)), _p(_i(dilation)), ci(int(transpose)), _p(outi), _p(pairs), _p(num))
    return outi[:m].copy(), pairs, num


def compact(pairs, num):
    assert all((pairs[k, :, num[k]:] == -1).all() for k in range(len(num)))  # the reference's own -1 fill
    pin = np.concatenate([pairs[k, 0, :num[k]] for k in range(len(num))])
    pout = np.concatenate([pairs[k, 1, :num[k]] for k in range(len(num))])
    return pin.astype(np.int32), pout.astype(np.int32)


def out_shape_conv(shape, k, s, p, d):
    return [(shape[i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(3)]  # spconv/ops.py:21-32


def out_shape_deconv(shape, k, s, p, d, op):
    return [(shape[i] - 1) * s[i] - 2 * p[i] + k[i] + op[i] for i in range(3)]  # spconv/ops.py:35-43


def benchmark_grid_indices(num_grids, points_per_grid, half_extent, cells, seed):
    """Voxel rows of bench.py's generator (occ_encoder.synthetic_object_grids: points uniform in the
    object box) in the order the product's front end emits them: sorted unique (b,z,y,x)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    xyz = ((torch.rand(num_grids * points_per_grid, 3, generator=g) * 2 - 1) * half_extent).numpy()
    vs = 2 * half_extent / np.asarray(cells[::-1], np.float32)  # x,y,z voxel size
    c = np.floor((xyz + half_extent) / vs).astype(np.int32).clip(0, np.asarray(cells[::-1]) - 1)
    b = np.repeat(np.arange(num_grids, dtype=np.int32), points_per_grid)
    rows = np.stack([b, c[:, 2], c[:, 1], c[:, 0]], 1)
    return np.unique(rows, axis=0).astype(np.int32)


def main():
    lib = load_ref()
    rng = np.random.default_rng(0)
    out = {}

    def put_subm(name, idx, batch, shape, ksize=(3, 3, 3), dilation=(1, 1, 1)):
        pairs, num = ref_subm(lib, idx, batch, shape, ksize, dilation)
        pin, pout = compact(pairs, num)
        out.update({f'{name}_indices': idx, f'{name}_batch': np.int32(batch), f'{name}_shape': _i(shape),
                    f'{name}_ksize': _i(ksize), f'{name}_dilation': _i(dilation), f'{name}_num': num,
                    f'{name}_pairs_in': pin, f'{name}_pairs_out': pout})
        print(name, 'N', len(idx), 'pairs', int(num.sum()))

    def put_conv(name, idx, batch, shape, ksize, stride, padding, dilation, transpose, out_padding=(0, 0, 0)):
        oshape = (out_shape_deconv(shape, ksize, stride, padding, dilation, out_padding) if transpose
                  else out_shape_conv(shape, ksize, stride, padding, dilation))
        outi, pairs, num = ref_conv(lib, idx, batch, oshape, ksize, stride, padding, dilation, transpose)
        pin, pout = compact(pairs, num)
        out.update({f'{name}_indices': idx, f'{name}_batch': np.int32(batch), f'{name}_shape': _i(shape),
                    f'{name}_out_shape': _i(oshape), f'{name}_ksize': _i(ksize), f'{name}_stride': _i(stride),
                    f'{name}_padding': _i(padding), f'{name}_dilation': _i(dilation),
                    f'{name}_transpose': np.int32(transpose), f'{name}_num': num, f'{name}_pairs_in': pin,
                    f'{name}_pairs_out': pout, f'{name}_out_indices': outi})
        print(name, 'N', len(idx), 'out', len(outi), 'pairs', int(num.sum()))
        return outi, oshape

    # (1) two 40^3 object grids of the benchmark generator (configs[1]: 2000 points per grid, 0.2 m)
    g40 = benchmark_grid_indices(2, 2000, 4.0, (40, 40, 40), seed=0)
    put_subm('bench40', g40, 2, (40, 40, 40))
    # (2) one 80^3 grid at 0.1 m (configs[4] cell size), 8000 points
    g80 = benchmark_grid_indices(1, 8000, 4.0, (80, 80, 80), seed=1)
    put_subm('bench80', g80, 1, (80, 80, 80))
    # (3) dense-ish, non-cubic, rows in random order, batch 3 (the CPU functor's order depends on row order)
    mask = rng.random((3, 9, 11, 13)) < 0.35
    idx = np.argwhere(mask).astype(np.int32)
    idx = idx[rng.permutation(len(idx))]
    put_subm('shuffled', idx, 3, (9, 11, 13))
    put_subm('dilated', idx, 3, (9, 11, 13), (3, 3, 3), (2, 2, 2))
    put_subm('k133', idx, 3, (9, 11, 13), (1, 3, 3))
    # (4) regular sparse conv: 3^3 stride 2 pad 1 on one benchmark grid; 2^3 stride 2 pad 0; anisotropic
    one40 = g40[g40[:, 0] == 0]
    d_out, d_shape = put_conv('down_k3s2p1', one40, 1, (40, 40, 40), (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), 0)
    put_conv('down_k2s2p0', g40, 2, (40, 40, 40), (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), 0)
    put_conv('down_aniso', idx, 3, (9, 11, 13), (3, 1, 3), (2, 1, 1), (1, 0, 1), (1, 1, 1), 0)
    put_conv('s1p0', idx, 3, (9, 11, 13), (3, 3, 3), (1, 1, 1), (0, 0, 0), (1, 1, 1), 0)
    # (5) transposed conv from the stride-2 outputs back up (SparseConvTranspose3d, conv.py:355-378)
    put_conv('up_k3s2p1', d_out, 1, d_shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), 1, (1, 1, 1))
    put_conv('up_k2s2p0', idx, 3, (9, 11, 13), (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), 1)

    # (6) the reference's CPU gather / scatter-add functors driving the indiceConv host loop
    #     (spconv_ops.h:300-354: centre offset = dense mm, every other offset gather -> mm -> scatter-add)
    lib.ref_sparse_gather_f32.restype = None
    lib.ref_sparse_scatter_add_f32.restype = None
    pairs, num = ref_subm(lib, idx, 3, (9, 11, 13))
    n, cin, cout = len(idx), 6, 10
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = rng.standard_normal((27, cin, cout)).astype(np.float32)
    y = x @ w[13]
    for k in range(27):
        if k == 13 or num[k] == 0:
            continue
        h = int(num[k])
        buf = np.zeros((h, cin), np.float32)
        lib.ref_sparse_gather_f32(_p(buf), _p(x), i64(n), ci(cin), _p(np.ascontiguousarray(pairs[k, 0, :h])), ci(h))
        ob = np.ascontiguousarray(buf @ w[k])
        lib.ref_sparse_scatter_add_f32(_p(y), i64(n), ci(cout), _p(ob), _p(np.ascontiguousarray(pairs[k, 1, :h])), ci(h))
    out.update(functor_x=x, functor_w=w.reshape(3, 3, 3, cin, cout), functor_y=y)

    dst = os.path.join(HERE, '..', 'tests', 'golden', 'rulebook.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
