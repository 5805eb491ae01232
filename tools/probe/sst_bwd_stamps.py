"""Phase timings of window_attn_block_bwd_kernel (csrc/window_block.hip) from in-kernel wall-clock stamps of every
workgroup's SECOND tile.  Builds its own copy of the library with -DOCOCC_WB_STAMPS (the product carries no stamps) and
runs the kernel on synthetic tokens like tools/probe/sst_block_bench.py.  Run on the GPU box:
    python tools/probe/sst_bwd_stamps.py [tokens] [mean window population]"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(ROOT, 'objectcentricocccompletion_amd', 'csrc')
so = '/tmp/libococc_wb_stamps.so'
obj = '/tmp/window_block_stamps.o'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-DOCOCC_WB_STAMPS', '-c',
                os.path.join(csrc, 'window_block.hip'), '-o', obj], check=True)
others = [o for o in glob.glob(os.path.join(csrc, 'build', '*.o')) if not o.endswith('window_block.o')]
subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', obj] + others + ['-o', so], check=True)
os.environ['OCOCC_LIB_PATH'] = so
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from objectcentricocccompletion_amd import _lib as L  # noqa: E402
from objectcentricocccompletion_amd.sst import fused_block as fb  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 260000
mean = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
lens = torch.poisson(torch.full((int(V / mean * 1.2),), mean), generator=g).clamp(1, 30).int()
cs = torch.cumsum(lens, 0)
nW = int((cs <= V).sum())
lens = lens[:nW]
V = int(lens.sum())
T = 30
tok = torch.full((nW * T,), -1, dtype=torch.int32)
perm = torch.randperm(V, generator=g).int()
start = torch.cumsum(lens, 0) - lens
slot = torch.repeat_interleave(torch.arange(nW) * T, lens.long()) + (torch.arange(V) - torch.repeat_interleave(start, lens.long()))
tok[slot] = perm
plan = fb.TilePlan([(tok.to(dev), lens.to(dev), nW, T)], dev)
E, F, H = 128, 256, 8
x = torch.randn(V, E, generator=g).bfloat16().to(dev)
pos = torch.randn(V, E, generator=g).bfloat16().to(dev)
dy = (torch.randn(V, E, generator=g) * 0.1).bfloat16().to(dev)
P = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(dev)
in_w, in_b, out_w, out_b = P(3 * E, E), P(3 * E), P(E, E), P(E)
wqkv, wo = fb.linear_fragments([in_w, out_w])
wot, wqkvt = fb.linear_fragments([out_w.t(), in_w.t()])
bq, bo = in_b.float().contiguous(), out_b.float().contiguous()
gg1 = torch.ones(E, device=dev)
dx, dz, o = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
dqkv = torch.empty((V, 3 * E), dtype=torch.bfloat16, device=dev)
prow2 = int(L.lib.ococc_window_block_partial_rows(plan.num_tiles))
lnp2 = torch.empty((prow2, 2, E), dtype=torch.float32, device=dev)
stamps = torch.zeros((1024 * 16,), dtype=torch.int64, device=dev)
L.lib.ococc_wb_set_stamps.argtypes = [ctypes.c_void_p]
assert L.lib.ococc_wb_set_stamps(stamps.data_ptr()) == 0


SAVED = os.environ.get('SST_STAMPS_RECOMPUTE', '0') != '1'   # default: the backward reads the forward's attention output back
y1 = torch.empty_like(x)
lse = torch.empty((V, H), dtype=torch.float32, device=dev)
bb1 = torch.zeros(E, device=dev)
L.check(L.lib.ococc_window_attn_block_train_fwd_bf16(
    L.ptr(x), L.ptr(pos), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, H, L.ptr(wqkv), L.ptr(bq), L.ptr(wo),
    L.ptr(bo), L.ptr(gg1), L.ptr(bb1), 1e-5, L.ptr(y1), L.ptr(o), L.ptr(lse), L.stream()), 'train fwd')


def run():
    if SAVED:
        L.check(L.lib.ococc_window_attn_block_bwd_saved_bf16(
            L.ptr(x), L.ptr(pos), L.ptr(dy), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, H, L.ptr(wqkv), L.ptr(bq),
            L.ptr(wo), L.ptr(bo), L.ptr(gg1), 1e-5, L.ptr(wot), L.ptr(wqkvt), L.ptr(o), L.ptr(lse), L.ptr(dx), L.ptr(dqkv),
            L.ptr(dz), L.ptr(lnp2), L.stream()), 'bwd saved')
    else:
        L.check(L.lib.ococc_window_attn_block_bwd_bf16(
            L.ptr(x), L.ptr(pos), L.ptr(dy), L.ptr(plan.rows), L.ptr(plan.span), plan.num_tiles, E, H, L.ptr(wqkv), L.ptr(bq),
            L.ptr(wo), L.ptr(bo), L.ptr(gg1), 1e-5, L.ptr(wot), L.ptr(wqkvt), L.ptr(dx), L.ptr(dqkv), L.ptr(dz), L.ptr(o),
            L.ptr(lnp2), L.stream()), 'bwd')


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record()
torch.cuda.synchronize()
print(f'{V} tokens, {plan.num_tiles} tiles: window_attn_block_bwd {a.elapsed_time(b) / 10 * 1e3:.1f} us per call')
stamps.zero_()
run()
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(-1, 16)
st = st[st[:, 0] > 0]
names = ['front: x + pos, Q|K|V', 'attention forward (or: kept o, lse -> LDS)', 'o stored, out-projection', 'LayerNorm fwd + bwd', 'dz1 stored, dO GEMM',
         'attention bwd pass 1 (dQ, delta)', 'pass 2 (dK, dV)', 'dQ|dK|dV to LDS + stored', 'dx GEMM', 'dx stored']
print(f'{len(st)} workgroups stamped their second tile; tile time {(st[:, 10] - st[:, 0]).mean() / 100:.2f} us')
for j, nme in enumerate(names):
    d = (st[:, j + 1] - st[:, j]) / 100.0
    print(f'  {nme:36s} {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})')

# inside the front (stamps of the workgroup's LAST tile): start -> x stashed + barrier -> V GEMM + meta + barrier -> x + pos,
# next tile's fetches + barrier -> (Q | K GEMM, stores, barrier: the rest of the front)
if st.shape[1] > 14 and (st[:, 11] > 0).all():
    for nme, j in (('x -> LDS, barrier', 11), ('next meta, V GEMM, V -> LDS, barrier', 12), ('x + pos -> LDS, next fetches, barrier', 13)):
        d = (st[:, j + 1] - st[:, j]) / 100.0
        print(f'    front / {nme:44s} {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})')
