"""numpy front end of the CPU oracle (oracle/ococc_oracle.c) plus pure-numpy /
torch-CPU restatements of the floating point stages.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by objectcentricocccompletion_amd/.
Each function names the reference file:line whose behaviour it restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libococc_oracle.so')


def build(verbose=False):
    """Compile the C restatement (and, where /root/reference exists, oracle/_ref)."""
    out = subprocess.run(['make', '-C', _HERE, 'all'], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError('oracle build failed:\n' + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


def _lib():
    if not os.path.exists(_SO):
        subprocess.run(['make', '-C', _HERE, '_build/libococc_oracle.so'], check=True,
                       capture_output=True)
    lib = ctypes.CDLL(_SO)
    lib.oracle_hard_voxelize.restype = ctypes.c_int
    for name in ('oracle_subm_rulebook', 'oracle_conv_rulebook', 'oracle_unique_rows', 'oracle_point_pool',
                 'oracle_deconv_rulebook'):
        getattr(lib, name).restype = ctypes.c_int64
    return lib


_L = None


def L():
    global _L
    if _L is None:
        _L = _lib()
    return _L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


i64 = ctypes.c_int64
ci = ctypes.c_int


def dynamic_voxelize(points, voxel_size, coors_range):
    """voxelization_cpu.cpp:8-41,145-169 -> coors [N,3] int32 (z,y,x)."""
    points = _f(points)
    coors = np.zeros((points.shape[0], 3), np.int32)
    L().oracle_dynamic_voxelize(_p(points), i64(points.shape[0]), ci(points.shape[1]),
                                _p(_f(voxel_size)), _p(_f(coors_range)), _p(coors))
    return coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """voxelization_cpu.cpp:44-143 -> (voxels, coors, num_points_per_voxel) sliced to voxel_num."""
    points = _f(points)
    nf = points.shape[1]
    voxels = np.zeros((max_voxels, max_points, nf), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    npv = np.zeros((max_voxels,), np.int32)
    vn = L().oracle_hard_voxelize(_p(points), i64(points.shape[0]), ci(nf), _p(_f(voxel_size)),
                                  _p(_f(coors_range)), ci(max_points), ci(max_voxels), _p(voxels),
                                  _p(coors), _p(npv))
    return voxels[:vn], coors[:vn], npv[:vn]


def subm_rulebook(indices, batch_size, spatial_shape, ksize=(3, 3, 3), dilation=(1, 1, 1)):
    """geometry.h:247-297 -> (indice_pairs [K,2,N] int32 with -1 fill, indice_num [K])."""
    indices = _i(indices)
    n = indices.shape[0]
    kvol = int(np.prod(ksize))
    pairs = np.full((kvol, 2, n), -1, np.int32)
    num = np.zeros((kvol,), np.int32)
    L().oracle_subm_rulebook(_p(indices), i64(n), ci(batch_size), _p(_i(spatial_shape)),
                             _p(_i(ksize)), _p(_i(dilation)), _p(pairs), _p(num))
    return pairs, num


def conv_rulebook(indices, batch_size, out_shape, ksize, stride, padding, dilation, transpose=False):
    """geometry.h:144-193 (or :195-245 transposed) -> (out_indices [M,4], indice_pairs [K,2,N], indice_num [K])."""
    indices = _i(indices)
    n = indices.shape[0]
    kvol = int(np.prod(ksize))
    pairs = np.full((kvol, 2, n), -1, np.int32)
    num = np.zeros((kvol,), np.int32)
    outi = np.zeros((max(n * kvol, 1), 4), np.int32)
    fn = L().oracle_deconv_rulebook if transpose else L().oracle_conv_rulebook
    m = fn(_p(indices), i64(n), ci(batch_size), _p(_i(out_shape)),
                                 _p(_i(ksize)), _p(_i(stride)), _p(_i(padding)), _p(_i(dilation)),
                                 _p(outi), _p(pairs), _p(num))
    return outi[:m].copy(), pairs, num


def indice_conv(features, filters, pairs, num, n_out, inverse=False, subm=False):
    """spconv_ops.h:260-361; filters [kD,kH,kW,Cin,Cout] or [K,Cin,Cout] -> out [n_out,Cout] f32."""
    features = _f(features)
    cin, cout = filters.shape[-2], filters.shape[-1]
    filters = _f(filters).reshape(-1, cin, cout)
    pairs, num = _i(pairs), _i(num)
    out = np.zeros((n_out, cout), np.float32)
    L().oracle_indice_conv(_p(features), i64(features.shape[0]), ci(cin), _p(filters),
                           ci(filters.shape[0]), ci(cout), _p(pairs), _p(num), i64(pairs.shape[2]),
                           i64(n_out), ci(int(inverse)), ci(int(subm)), _p(out))
    return out


def indice_conv_backward(features, filters, dout, pairs, num, inverse=False, subm=False):
    """spconv_ops.h:363-456 -> (din [n_in,Cin], dfilters like filters)."""
    features, dout = _f(features), _f(dout)
    shape = filters.shape
    cin, cout = shape[-2], shape[-1]
    filters = _f(filters).reshape(-1, cin, cout)
    pairs, num = _i(pairs), _i(num)
    din = np.zeros_like(features)
    dfil = np.zeros_like(filters)
    L().oracle_indice_conv_backward(_p(features), i64(features.shape[0]), ci(cin), _p(filters),
                                    ci(filters.shape[0]), ci(cout), _p(dout), i64(dout.shape[0]),
                                    _p(pairs), _p(num), i64(pairs.shape[2]), ci(int(inverse)),
                                    ci(int(subm)), _p(din), _p(dfil))
    return din, dfil.reshape(shape)


_RED = {'sum': 0, 'mean': 1, 'avg': 1, 'max': 2}


def segment_reduce(feats, inv, num_segments, mode):
    """scatter_points_cuda.cu:81-103 / torch_scatter semantics (sst_ops.py:171-174).
    -> (out [G,C] f32, counts [G] i32, arg [G,C] i32 (max only, smallest row index))."""
    feats, inv = _f(feats), _i(inv)
    n, c = feats.shape
    out = np.zeros((num_segments, c), np.float32)
    counts = np.zeros((num_segments,), np.int32)
    arg = np.zeros((num_segments, c), np.int32)
    L().oracle_segment_reduce(_p(feats), _p(inv), i64(n), ci(c), ci(_RED[mode]), i64(num_segments),
                              _p(out), _p(counts), _p(arg))
    return out, counts, arg


def unique_rows(coors):
    """at::unique_dim(sorted, inverse, counts) with negative rows dropped
    (scatter_points_cuda.cu:199-210) -> (out_coors [U,ndim], inv [N], counts [U])."""
    coors = _i(coors)
    if coors.ndim == 1:
        coors = coors[:, None]
    n, ndim = coors.shape
    outc = np.zeros((max(n, 1), ndim), np.int32)
    inv = np.zeros((n,), np.int32)
    counts = np.zeros((max(n, 1),), np.int32)
    u = L().oracle_unique_rows(_p(coors), i64(n), ci(ndim), _p(outc), _p(inv), _p(counts))
    return outc[:u].copy(), inv, counts[:u].copy()


def dynamic_scatter(feats, coors, mode):
    """DynamicScatter forward (scatter_points_cuda.cu:183-234) and the brute-force recipe of
    tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:56-65."""
    outc, inv, counts = unique_rows(coors)
    out, _, _ = segment_reduce(feats, inv, outc.shape[0], mode)
    return out, outc, inv, counts


def point_pool(rois, roi_key, pts, pts_key, extra_wlh, max_inbox_point, max_all_pts):
    """TorchEx dynamic_point_pool_mixed contract (dynamic_point_pool_op.py:63-113), rows sorted by
    (roi, point) -> (pts_idx [M] i64, roi_idx [M] i64, feats [M,13] f32, roi_counts [R] i32)."""
    rois, pts = _f(rois), _f(pts)
    roi_key, pts_key = _i(roi_key), _i(pts_key)
    R, N = rois.shape[0], pts.shape[0]
    op = np.full((max_all_pts,), -1, np.int64)
    orr = np.full((max_all_pts,), -1, np.int64)
    fe = np.zeros((max_all_pts, 13), np.float32)
    rc = np.zeros((R,), np.int32)
    m = L().oracle_point_pool(_p(rois), _p(roi_key), i64(R), _p(pts), _p(pts_key), i64(N),
                              _p(_f(extra_wlh)), ci(max_inbox_point), i64(max_all_pts), _p(op), _p(orr),
                              _p(fe), _p(rc))
    return op[:m].copy(), orr[:m].copy(), fe[:m].copy(), rc


def aligned_iou3d(boxes1, boxes2):
    """lidar_box3d.py:404-448 -> iou [n] f32 (i-th box with i-th box)."""
    b1, b2 = _f(boxes1), _f(boxes2)
    out = np.zeros((b1.shape[0],), np.float32)
    L().oracle_aligned_iou3d(_p(b1), _p(b2), i64(b1.shape[0]), _p(out))
    return out


def bev_overlap_1to1(xyxyr1, xyxyr2):
    """TorchEx boxes_overlap_1to1 contract (lidar_box3d.py:429-434) -> BEV intersection area [n] f32."""
    b1, b2 = _f(xyxyr1), _f(xyxyr2)
    out = np.zeros((b1.shape[0],), np.float32)
    L().oracle_bev_overlap_1to1(_p(b1), _p(b2), i64(b1.shape[0]), _p(out))
    return out


# ------------------------------------------------------------------ floating point helpers
def bf16_round(a):
    """round-to-nearest-even float32 -> bfloat16 -> float32 (numpy)."""
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def layernorm_act(x, gamma, beta, eps, act):
    """nn.LayerNorm + nn.GELU() (exact erf), the pair make_sparse_convmodule /
    build_mlp append (sparse_block.py:216-289, sst_ops.py:333-360); float64 inside."""
    from math import sqrt
    from scipy.special import erf
    x = np.asarray(x, np.float64)
    mu = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    z = (x - mu) / np.sqrt(var + eps) * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
    if act:
        z = 0.5 * z * (1.0 + erf(z / sqrt(2.0)))
    return z


def indice_maxpool(features, pairs, num, n_out):
    """src/maxpool.cc:9-27 with the zero-initialised output of pool_ops.h:34 -> out [n_out, C] f32."""
    features = _f(features)
    pairs, num = _i(pairs), _i(num)
    out = np.zeros((n_out, features.shape[1]), np.float32)
    for k in range(pairs.shape[0]):
        for r in range(int(num[k])):
            i, o = int(pairs[k, 0, r]), int(pairs[k, 1, r])
            out[o] = np.where(out[o] < features[i], features[i], out[o])
    return out


def indice_maxpool_backward(features, out_features, dout, pairs, num):
    """src/maxpool.cc:31-53 (din zero-initialised, pool_ops.h:71) -> din [n_in, C] f32."""
    features, out_features, dout = _f(features), _f(out_features), _f(dout)
    pairs, num = _i(pairs), _i(num)
    din = np.zeros_like(features)
    for k in range(pairs.shape[0]):
        for r in range(int(num[k])):
            i, o = int(pairs[k, 0, r]), int(pairs[k, 1, r])
            din[i] += np.where(out_features[o] == features[i], dout[o], np.float32(0))
    return din
