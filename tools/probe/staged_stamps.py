"""Round-4 experiment record: the "staged" 64 -> 128 convolution (tools/probe/sparse_conv_staged.hip.txt -- every operand
through LDS, gathers compacted once per launch, two waves per 64-row group = 16 waves per CU, item loop fully unrolled).
Correct (equal to the product kernel to f32 rounding on sparse and dense tables, deterministic).  Measured at the benchmark
size (128 k rows, eager launches, interleaved rounds in one process): 40-41 us against the product's streamed-weights
kernel at 44 us.  What the versions taught:
  * compaction inside the loop: 56-60 us -- ~120 scalar + ~70 vector instructions per item and wave saturate the SIMDs'
    issue at 4 waves per SIMD (one scalar and one vector issue per 4 cycles), LDS round trips cost 300-400 cycles;
  * tables once per launch + unrolled items (this file): the loop runs 1.5-1.8 k cycles per item although a wave issues
    ~80 instructions: what is left is the matrix pipe (768 cycles per SIMD and item on average at 6 x over-issue) and its
    imbalance at the per-item barrier; issuing all LDS reads of an item up front changed nothing;
  * the prologue (table build: ~35 instructions per column and wave, 14 columns) is 20 k of a wave's 65-70 k cycles.
This script builds the archived translation unit (optionally with in-kernel shader-clock stamps: 'stamps' = per phase,
which drains the LDS queue at every stamp -- read shares only; 'coarse' = prologue / loop / epilogue), checks it against
the product kernel and times both.
Run on the GPU box: python tools/probe/staged_stamps.py [row_groups [stamps|coarse]]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'objectcentricocccompletion_amd', 'csrc')
so = '/tmp/libstaged_stamps.so'
here = os.path.dirname(os.path.abspath(__file__))
STAMPS = len(sys.argv) > 2 and sys.argv[2] in ('stamps', 'coarse')
COARSE = STAMPS and sys.argv[2] == 'coarse'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-Wno-inline-asm']
               + (['-DOCOCC_STAGED_STAMPS=' + ('2' if COARSE else '1')] if STAMPS else []) + ['-I', csrc, '-x', 'hip', os.path.join(here, 'sparse_conv_staged.hip.txt'),
                os.path.join(csrc, 'capi.hip'), '-o', so], check=True)
lib = ctypes.CDLL(so)
from objectcentricocccompletion_amd import _lib as L  # noqa: E402
from objectcentricocccompletion_amd.spconv import ops  # noqa: E402

rg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
B = 64
cells = torch.stack([torch.randperm(64000, generator=g)[:2000].sort().values + b * 64000 for b in range(B)]).flatten()
idx = torch.stack([cells // 64000, (cells // 1600) % 40, (cells // 40) % 40, cells % 40], 1).int().to(dev)
n = idx.shape[0]
_, pairs, num = ops.get_indice_pairs(idx, B, [40, 40, 40], 3, subm=True)
table = pairs._ococc.tables[(False, 'fwd')][0]
x = torch.randn(n, 64, generator=g).to(dev).bfloat16()
w = (torch.randn(3, 3, 3, 64, 128, generator=g) * 0.05).to(dev)
# the kernel's operand: fragment-major (prepare mode + 4) with the output channels interleaved so that A-row i of column
# block b carries channel (b >> 1) * 32 + (i >> 2) * 8 + (b & 1) * 4 + (i & 3) (16-byte stores in the epilogue)
perm = torch.tensor([(b >> 1) * 32 + (i >> 2) * 8 + (b & 1) * 4 + (i & 3) for b in range(8) for i in range(16)], device=dev)
wn = ops._prep_weights(w[..., perm].contiguous(), 4, 64, 128)
out = torch.empty((n, 128), dtype=torch.bfloat16, device=dev)
nw = rg * 2
rows = rg * 64
grid = (-(-n // rows) + 7) // 8 * 8
stamps = torch.zeros((grid * nw * 8,), dtype=torch.int64, device=dev)
if STAMPS:
    lib.ococc_staged_set_stamps.argtypes = [ctypes.c_void_p]
    assert lib.ococc_staged_set_stamps(stamps.data_ptr()) == 0
lib.ococc_sparse_conv_staged_probe(rg)
vp = ctypes.c_void_p
lib.ococc_sparse_conv_staged_bf16.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32, vp,
                                              ctypes.c_int32, ctypes.c_int64, vp, vp, ctypes.c_int32, vp]


def run():
    rc = lib.ococc_sparse_conv_staged_bf16(x.data_ptr(), n, 64, wn.data_ptr(), 27, 128, table.data_ptr(), 13, n, None,
                                           out.data_ptr(), L.BF16, None)
    assert rc == 0, rc


def staged(xb, wnb, tab, nrows, out_dtype, dense_k=13):
    o = torch.empty((nrows, 128), dtype=out_dtype, device=dev)
    rc = lib.ococc_sparse_conv_staged_bf16(xb.data_ptr(), xb.size(0), 64, wnb.data_ptr(), 27, 128, tab.data_ptr(), dense_k, nrows,
                                           None, o.data_ptr(), L.dtype_code(out_dtype), None)
    assert rc == 0, rc
    return o


# parity on small grids first: sparse, dense (groups that overflow a stage), ragged tails; f32 and bf16 outputs
import numpy as np  # noqa: E402
for dens in (0.02, 0.05, 0.3, 0.9):
    rng = np.random.default_rng(int(dens * 100))
    shape = (14, 15, 16)
    cells = np.concatenate([np.sort(rng.choice(14 * 15 * 16, int(dens * 14 * 15 * 16), replace=False)) + b * 14 * 15 * 16 for b in range(3)])
    ii = torch.from_numpy(np.stack([cells // (14 * 15 * 16), (cells // (15 * 16)) % 14, (cells // 16) % 15, cells % 16], 1).astype(np.int32)).to(dev)
    m = ii.shape[0]
    _, pp, nn = ops.get_indice_pairs(ii, 3, list(shape), 3, subm=True)
    tt = pp._ococc.tables[(False, 'fwd')][0]
    xx = torch.randn(m, 64, generator=g).to(dev).bfloat16()
    ref32 = ops.indice_conv(xx.float(), w.bfloat16().float(), pp, nn, m, False, True)
    for dk in (13, -1):
        got32 = staged(xx, wn, tt, m, torch.float32, dk)
        gotbf = staged(xx, wn, tt, m, torch.bfloat16, dk)
        err = float((got32 - ref32).abs().max() / ref32.abs().max())
        again = staged(xx, wn, tt, m, torch.float32, dk)
        print(f'density {dens}: rows {m}, dense_k {dk}: max rel diff to the product kernel {err:.2e}, bf16 = RNE(f32): '
              f'{bool(torch.equal(gotbf, got32.bfloat16()))}, deterministic: {bool(torch.equal(got32, again))}')
        assert err < 1e-5

for _ in range(5):
    run()
torch.cuda.synchronize()
ref = ops.indice_conv(x, w, pairs, num, n, False, True)
print('benchmark size: max |staged - product kernel|:', float((out.float() - ref.float()).abs().max()))
times = {'product stream kernel': [], f'staged rg={rg}': []}
for r in range(7):
    for name, fn in (('product stream kernel', lambda: ops.indice_conv(x, w, pairs, num, n, False, True)), (f'staged rg={rg}', run)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name, ts in times.items():
    ts = sorted(ts)
    print(f'{name:24s} median {ts[len(ts) // 2]:7.2f} us  min {ts[0]:7.2f} us (eager launches incl. host gaps' + (', stamped build)' if STAMPS else ')'))
if STAMPS:
    st = stamps.view(grid, nw, 8).cpu().double()
    live = st[:, :, 6].sum(1) > 0
    st = st[live]
    names = ['prologue', 'the whole item loop' if COARSE else 'item: next item + DMA issue', 'item: products', 'item: wait for the DMAs', 'item: barrier', 'epilogue (stores)']
    for role, sel in (('stagers (even waves)', st[:, 0::2]), ('partners (odd waves)', st[:, 1::2])):
        tot = sel[:, :, :6].sum(2).mean()
        print(f'{role}: {sel[:, :, 6].mean():.1f} items, {tot:.0f} cycles per wave')
        for i, nm in enumerate(names):
            v = sel[:, :, i].mean()
            print(f'   {nm:32s} {v:9.0f} cycles  {100 * v / tot:5.1f} %' + (f'   ({v / sel[:, :, 6].mean():.0f} per item)' if 1 <= i <= 4 else ''))
