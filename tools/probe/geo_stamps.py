"""Phase timings of the fused geometry kernels (csrc/grid_geometry.hip) from in-kernel wall-clock stamps.
Builds its own copy of the translation unit with -DOCOCC_GEO_STAMPS (the product library carries no stamps) and calls
it through ctypes on the benchmark batch.  Run on the GPU box: python tools/probe/geo_stamps.py [slices]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'objectcentricocccompletion_amd', 'csrc')
so = '/tmp/libgeo_stamps.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-DOCOCC_GEO_STAMPS',
                os.path.join(csrc, 'grid_geometry.hip'), os.path.join(csrc, 'grid_unique.hip'), os.path.join(csrc, 'capi.hip'),
                os.path.join(csrc, 'sparse_conv_sorted.hip'),   # (the geometry call launches the order's placing pass)
                '-o', so], check=True)
lib = ctypes.CDLL(so)
from objectcentricocccompletion_amd import _lib as L  # noqa: E402  (argument helpers only)
from objectcentricocccompletion_amd.occ_encoder import synthetic_object_grids  # noqa: E402

slices = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
B, P = 64, 2000
xyz, feats, bidx = synthetic_object_grids(B, P, seed=0, device=dev)
n, c, cap = B * P, 16, B * P
I3, F3, F6 = ctypes.c_int32 * 3, ctypes.c_float * 3, ctypes.c_float * 6
lib.ococc_object_grid_geometry_workspace_bytes.restype = ctypes.c_int64
lib.ococc_object_grid_geometry_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32, I3, ctypes.c_int32]
nbytes = lib.ococc_object_grid_geometry_workspace_bytes(n, B, I3(40, 40, 40), slices)
T = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
ws, coors, inv, counts = T((nbytes,), torch.uint8), T((cap, 4), torch.int32), T((n,), torch.int32), T((cap,), torch.int32)
out, out16, meta = T((cap, c), torch.float32), T((cap, c), torch.bfloat16), T((2,), torch.int32)
nbr, mask, pairs, num = T((27, cap), torch.int32), T(((cap + 15) // 16,), torch.int32), T((27, 2, cap), torch.int32), T((27,), torch.int32)
nblocks = B * slices + (cap + 1023) // 1024
stamps = torch.zeros((max(nblocks, B) * 16,), dtype=torch.int64, device=dev)
lib.ococc_geo_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.ococc_geo_set_stamps(stamps.data_ptr()) == 0
vp = ctypes.c_void_p
lib.ococc_object_grid_geometry_f32.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int64, vp, ctypes.c_int32, F3, F6, ctypes.c_int32, I3,
                                               ctypes.c_int32, vp, ctypes.c_int64] + [vp] * 11 + [ctypes.c_int64, vp]


def run():
    rc = lib.ococc_object_grid_geometry_f32(xyz.data_ptr(), 3, bidx.data_ptr(), n, feats.data_ptr(), c, F3(0.2, 0.2, 0.2),
                                            F6(-4, -4, -4, 4, 4, 4), B, I3(40, 40, 40), slices, coors.data_ptr(), cap,
                                            inv.data_ptr(), counts.data_ptr(), out.data_ptr(), out16.data_ptr(),
                                            meta.data_ptr(), meta.data_ptr() + 4, nbr.data_ptr(), mask.data_ptr(),
                                            pairs.data_ptr(), num.data_ptr(), ws.data_ptr(), nbytes, None)
    assert rc == 0, rc


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print('three launches, us per call:', round(e0.elapsed_time(e1) / 20 * 1e3, 1), ' voxels', int(meta[0]))
st = stamps.cpu().numpy().reshape(-1, 16).astype(np.float64) / 100.0   # s_memrealtime ticks at 100 MHz -> us
A = st[:B * min(slices, 4)]
names_a = ['zero+segments', 'point pass', 'scan+bitmap out', 'neighbour counts']
print('kernel A (median over workgroups, us):', {nm: round(float(np.median(A[:, i + 1] - A[:, i])), 2) for i, nm in enumerate(names_a)},
      'whole', round(float(A[:, 4].max() - A[:, 0].min()), 2))
Bk = st[:B * slices]
names_b = ['load bitmap+scan', 'row loop', 'pass1 inv+first', 'pass2 dups', 'pass3 means']
print('kernel B (median over workgroups, us):', {nm: round(float(np.median(Bk[:, i + 6] - Bk[:, i + 5])), 2) for i, nm in enumerate(names_b)},
      'whole', round(float(Bk[:, 10].max() - Bk[:, 5].min()), 2))
print('row loop detail (last round): cells', round(float(np.median(Bk[:, 11] - Bk[:, 6])), 2), 'pass 0 (table, counts)', round(float(np.median(Bk[:, 12] - Bk[:, 11])), 2), 'masks', round(float(np.median(Bk[:, 13] - Bk[:, 12])), 2), 'prefix', round(float(np.median(Bk[:, 14] - Bk[:, 13])), 2), 'pass 1 (pairs)', round(float(np.median(Bk[:, 7] - Bk[:, 14])), 2))
