"""Phase timings of the compact-then-multiply sub-manifold convolution (csrc/sparse_conv_tile.hip) from in-kernel
wall-clock stamps.  Builds its own copy of the translation unit with -DOCOCC_TILE_STAMPS (the product library carries no
stamps).  Run on the GPU box: python tools/probe/tile_stamps.py <kd> <ncols> [lnbwd]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'objectcentricocccompletion_amd', 'csrc')
so = '/tmp/libtile_stamps.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-DOCOCC_TILE_STAMPS']
               + os.environ.get('TILE_DEFS', '').split() +
               [os.path.join(csrc, 'sparse_conv_tile.hip'), os.path.join(csrc, 'capi.hip'), '-o', so], check=True)
lib = ctypes.CDLL(so)
from objectcentricocccompletion_amd import _lib as L  # noqa: E402
from objectcentricocccompletion_amd.spconv import ops  # noqa: E402

kd, nc = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
B = 64
cells = torch.stack([torch.randperm(64000, generator=g)[:1970].sort().values + b * 64000 for b in range(B)]).flatten()
idx = torch.stack([cells // 64000, (cells // 1600) % 40, (cells // 40) % 40, cells % 40], 1).int().to(dev)
n = idx.shape[0]
_, pairs, num = ops.get_indice_pairs(idx, B, [40, 40, 40], 3, subm=True)
table = pairs._ococc.tables[(False, 'fwd')][0]
x = torch.randn(n, kd, generator=g).to(dev).bfloat16()
w = (torch.randn(3, 3, 3, kd, nc, generator=g) * 0.05).to(dev)
wn = ops._prep_weights(w, 4, kd, nc)   # fragment-major forward operand
out = torch.empty((n, nc), dtype=torch.bfloat16, device=dev)
stamps = torch.zeros((2048 * 16,), dtype=torch.int64, device=dev)
lib.ococc_tile_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.ococc_tile_set_stamps(stamps.data_ptr()) == 0
vp = ctypes.c_void_p
lib.ococc_sparse_conv_tile_bf16.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int32,
                                            ctypes.c_int64, vp, vp, ctypes.c_int32, vp]


LNB = len(sys.argv) > 3 and sys.argv[3] == 'lnbwd'   # the input-gradient instantiation with the LayerNorm-backward epilogue
if LNB:
    conv_out = torch.randn(n, nc, generator=g).to(dev).bfloat16()
    mu = conv_out.float().mean(1)
    stats = torch.stack([mu, 1.0 / torch.sqrt(conv_out.float().var(1, unbiased=False) + 1e-3)], 1).contiguous()
    gamma, beta = torch.ones(nc, device=dev), torch.zeros(nc, device=dev)
    lib.ococc_sparse_conv_tile_lnbwd_partial_rows.restype = ctypes.c_int64
    lib.ococc_sparse_conv_tile_lnbwd_partial_rows.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    prows = lib.ococc_sparse_conv_tile_lnbwd_partial_rows(n, kd, nc)
    partials = torch.empty((prows, 2 * nc), dtype=torch.float32, device=dev)
    lib.ococc_sparse_conv_tile_lnbwd_bf16.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32, vp,
                                                      ctypes.c_int32, ctypes.c_int64, vp, vp, vp, vp, ctypes.c_int32, vp, vp,
                                                      ctypes.c_int64, vp]


def run():
    if LNB:
        rc = lib.ococc_sparse_conv_tile_lnbwd_bf16(x.data_ptr(), n, kd, wn.data_ptr(), 27, nc, table.data_ptr(), 13, n,
                                                   conv_out.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1,
                                                   out.data_ptr(), partials.data_ptr(), prows, None)
    else:
        rc = lib.ococc_sparse_conv_tile_bf16(x.data_ptr(), n, kd, wn.data_ptr(), 27, nc, table.data_ptr(), 13, n, None,
                                             out.data_ptr(), L.BF16, None)
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
print(f'tile conv {kd} -> {nc}: us per launch', round(e0.elapsed_time(e1) / 20 * 1e3, 1))
# the contraction itself, f32 on the device: out[o] = sum_k x[table[k][o]] @ w[k]
tb = table.view(27, n).long()
ref = torch.zeros(n, nc, device=dev)
wf = w.view(27, kd, nc).bfloat16().float()
for k in range(27):
    m = tb[k] >= 0
    ref[m] += x[tb[k][m]].float() @ wf[k]
if not LNB:
    err = (out.float() - ref).abs().max().item()
    print('max |out - f32 reference|', err, 'of', ref.abs().max().item(), '(bf16 output rounding: 2^-9 relative)')
st = stamps.cpu().numpy().reshape(-1, 16).astype(np.float64) / 100.0
st = st[st[:, 0] > 0]
if len(st) == 0:
    sys.exit(0)
if nc >= 128:
    names = ['zero masks, dense rows requested', 'table columns + ranks', 'gathers issued + barrier', 'chunk 0: dense product', 'chunk 0: produce', 'chunk 0: barrier', 'chunk 0: pull', 'chunk 0: barrier 2', 'chunk 0 store + other chunks']
else:
    names = ['tile init', 'dense offset', 'pass0: table cols + ranks', 'pass0: gathers issued + w', 'pass0 j0: MFMA', 'pass0 j0: barrier wait',
         'pass0 j0: ordered adds', 'rest of the passes', 'epilogue']
print({nm: round(float(np.median(st[:, i + 1] - st[:, i])), 2) for i, nm in enumerate(names)})
print('workgroups', len(st), 'median lifetime', round(float(np.median(st[:, 9] - st[:, 0])), 2), 'first start -> last end', round(float(st[:, 9].max() - st[:, 0].min()), 2))
