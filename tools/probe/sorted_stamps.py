"""Phase timings of the neighbour-pattern-order convolution (csrc/sparse_conv_sorted.hip) from in-kernel wall-clock
stamps, per tile class.  Builds its own copy of the translation unit with -DOCOCC_SORTED_STAMPS (the product library
carries no stamps).  Run on the GPU box: python tools/probe/sorted_stamps.py [heavy,mid]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'objectcentricocccompletion_amd', 'csrc')
so = '/tmp/libsorted_stamps.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-DOCOCC_SORTED_STAMPS']
               + os.environ.get('SORTED_DEFS', '').split() +
               [os.path.join(csrc, 'sparse_conv_sorted.hip'), os.path.join(csrc, 'capi.hip'), '-o', so], check=True)
lib = ctypes.CDLL(so)
from objectcentricocccompletion_amd import _lib as L  # noqa: E402
from objectcentricocccompletion_amd.spconv import ops  # noqa: E402

hb, mb = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,8').split(',')]
kd, nc = 64, 128
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
B = 64
cells = torch.stack([torch.randperm(64000, generator=g)[:1970].sort().values + b * 64000 for b in range(B)]).flatten()
idx = torch.stack([cells // 64000, (cells // 1600) % 40, (cells // 40) % 40, cells % 40], 1).int().to(dev)
n = idx.shape[0]
_, pairs, num = ops.get_indice_pairs(idx, B, [40, 40, 40], 3, subm=True)
rb = pairs._ococc
table = rb.tables[(False, 'fwd')][0]
ops.SORTED_TILES = (hb, mb)
rec, hdr = ops.row_order(rb, table, n)
x = torch.randn(n, kd, generator=g).to(dev).bfloat16()
w = (torch.randn(3, 3, 3, kd, nc, generator=g) * 0.05).to(dev)
wn = ops._prep_weights(w, 0, kd, nc)
out = torch.empty((n, nc), dtype=torch.bfloat16, device=dev)
stamps = torch.zeros((4096 * 8,), dtype=torch.int64, device=dev)
lib.ococc_sorted_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.ococc_sorted_set_stamps(stamps.data_ptr()) == 0
vp = ctypes.c_void_p
lib.ococc_sparse_conv_sorted_bf16.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32, vp, vp, vp,
                                              ctypes.c_int64, vp, vp, ctypes.c_int32, vp]


def run():
    rc = lib.ococc_sparse_conv_sorted_bf16(x.data_ptr(), n, kd, wn.data_ptr(), 27, nc, table.data_ptr(), rec.data_ptr(),
                                           hdr.data_ptr(), n, None, out.data_ptr(), L.BF16, None)
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
print(f'sorted conv {kd} -> {nc} tiles {hb},{mb}: us per call', round(e0.elapsed_time(e1) / 20 * 1e3, 1), ' hdr', hdr.tolist())
ref = ops.indice_conv(x, w, pairs, num, n, False, True)
print('equal to the library result:', bool(torch.equal(ref, out)))
stamps.zero_()
run()
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(-1, 8)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
print(f'{len(st)} tiles; first start .. last end: {(st[:, 4].max() - t0) / 100:.1f} us')
its = st[:, 6]
for lo, hi in ((1, 1), (2, 3), (4, 6), (7, 12), (13, 20), (21, 32)):
    m = (its >= lo) & (its <= hi)
    if not m.any():
        continue
    s = st[m]
    ph = [(s[:, j + 1] - s[:, j]).mean() / 100 for j in range(4)]
    print(f'offsets {lo:2d}-{hi:2d}: {m.sum():4d} tiles  start at {((s[:, 0] - t0).mean()) / 100:6.1f} us   masks+table {ph[0]:5.1f}  '
          f'first operands {ph[1]:5.1f}  loop {ph[2]:5.1f} ({ph[2] / s[:, 6].mean():.2f} per offset)  stores issued {ph[3]:5.1f}   '
          f'end at {((s[:, 4] - t0).mean()) / 100:6.1f} (max {((s[:, 4] - t0).max()) / 100:6.1f})')
