"""Round-4 experiment record: the "staged" 64 -> 128 convolution (tools/probe/sparse_conv_staged.hip.txt -- every operand
through LDS, gathers compacted once per launch, two waves per 64-row group = 16 waves per CU).  Correct (oracle parity,
deterministic) but SLOWER than the streamed-weights kernel of the product (56 vs 42 us eager): at 16 waves per CU the
SIMDs' instruction issue saturates (4 waves x 26 % issue) and every LDS round trip of an item costs 300-400 cycles.
This script builds the archived translation unit (optionally with in-kernel shader-clock stamps summed per phase and wave:
read SHARES, not totals), checks it against the product kernel and times both.
Run on the GPU box: python tools/probe/staged_stamps.py [row_groups]"""
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
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-Wno-inline-asm',
                '-DOCOCC_STAGED_STAMPS', '-I', csrc, '-x', 'hip', os.path.join(here, 'sparse_conv_staged.hip.txt'),
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


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ref = ops.indice_conv(x, w, pairs, num, n, False, True)
print('max |staged - product kernel|:', float((out.float() - ref.float()).abs().max()), '(one bf16 step of the largest value: sums in another order)')
print(f'staged conv rg={rg} (stamped build): us per launch', round(e0.elapsed_time(e1) / 20 * 1e3, 1))
st = stamps.view(grid, nw, 8).cpu().double()
live = st[:, :, 6].sum(1) > 0
st = st[live]
names = ['prologue', 'item: next item + DMA issue', 'item: products', 'item: wait for the DMAs', 'item: barrier', 'epilogue (stores)']
for role, sel in (('stagers (even waves)', st[:, 0::2]), ('partners (odd waves)', st[:, 1::2])):
    tot = sel[:, :, :6].sum(2).mean()
    print(f'{role}: {sel[:, :, 6].mean():.1f} items, {tot:.0f} cycles per wave')
    for i, nm in enumerate(names):
        v = sel[:, :, i].mean()
        print(f'   {nm:32s} {v:9.0f} cycles  {100 * v / tot:5.1f} %' + (f'   ({v / sel[:, :, 6].mean():.0f} per item)' if 1 <= i <= 4 else ''))
