"""Where the HOST spends an SST training step (bench.py --workload sst's step): cProfile of 20 steps + the synchronising
call sites (torch sync-debug warnings).  Run on the GPU box: python tools/probe/sst_host_profile.py"""
import collections, cProfile, os, pstats, sys, time, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from objectcentricocccompletion_amd.occ_encoder import synthetic_object_grids
from objectcentricocccompletion_amd.optim import AdamW
from objectcentricocccompletion_amd.sst import sst_modules as sm
from objectcentricocccompletion_amd.voxel import dynamic_scatter, voxelization
from objectcentricocccompletion_amd.linear import Linear as TallLinear
dev = torch.device('cuda:0')
torch.manual_seed(0)
G, P = 32, 8200
shape = (64, 80, 80)
rng = [-4, -4, -3.2, 4, 4, 3.2]
drop = {0: dict(max_tokens=30, drop_range=(0, 30)), 1: dict(max_tokens=60, drop_range=(30, 60)),
        2: dict(max_tokens=100, drop_range=(60, 100000))}
inp = sm.SSTInputLayerV2(drop, (8, 8, 8), (80, 80, 64), shuffle_voxels=False, debug=False, mute=True).to(dev)
model = sm.SSTv2(d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2, dropout=0.0,
                 activation='gelu', num_attached_conv=0, to_bev=False, debug=False,
                 layer_cfg=dict(compute_dtype=torch.bfloat16)).to(dev).train()
embed = TallLinear(16, 128).to(dev)
params = list(model.parameters()) + list(embed.parameters())
opt = AdamW(params, lr=1e-4)
xyz, feats, bidx = synthetic_object_grids(G, P, seed=0, device=dev)
xyz[:, 2] *= 0.8
with torch.no_grad():
    zyx = voxelization(xyz, [0.1, 0.1, 0.1], rng, -1, -1)
    n_act = dynamic_scatter(feats, torch.cat([bidx.view(-1, 1).to(torch.int32), zyx], 1), 'mean', grid_shape=[G] + list(shape))[0].shape[0]
d_out = (torch.randn(n_act, 128, device=dev) / n_act).to(torch.bfloat16)
marks = {}


def step(timed=False):
    t = [time.perf_counter()]
    def mark():
        if timed:
            torch.cuda.synchronize()
            t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True)
    zyx = voxelization(xyz, [0.1, 0.1, 0.1], rng, -1, -1)
    coors = torch.cat([bidx.view(-1, 1).to(torch.int32), zyx], 1)
    vfeats, vcoors = dynamic_scatter(feats, coors, 'mean', grid_shape=[G] + list(shape))
    mark()
    info = inp(embed(vfeats), vcoors.long(), batch_size=G)
    mark()
    out = model(info)[0]['voxel_feats']
    mark()
    out.backward(d_out)
    mark()
    opt.step()
    mark()
    if timed:
        for name, a, b in zip(('voxelise+scatter', 'embed+input layer', 'blocks fwd', 'backward', 'optimizer'), t[:-1], t[1:]):
            marks[name] = marks.get(name, 0.0) + (b - a)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print('step %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
for _ in range(10):
    step(True)
print('phases with a synchronise after each (ms):', {k: round(v / 10 * 1e3, 3) for k, v in marks.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
sites = collections.Counter()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step()
torch.cuda.set_sync_debug_mode('default')
print('synchronising calls in one step:', len(w))
import traceback as tb
