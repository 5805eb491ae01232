"""Host synchronisations of one --workload sst training step, by call site (torch sync-debug warnings), and the host time of
the input layer / the backbone.  usage (GPU box): python tools/find_syncs_sst.py"""
import collections
import os
import sys
import time
import traceback
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectcentricocccompletion_amd.occ_encoder import synthetic_object_grids
from objectcentricocccompletion_amd.sst import sst_modules as sm
from objectcentricocccompletion_amd.voxel import dynamic_scatter, voxelization

dev = torch.device('cuda:0')
torch.manual_seed(0)
G, P = 32, 8200
shape = (64, 80, 80)
rng = [-4, -4, -3.2, 4, 4, 3.2]
drop = {0: dict(max_tokens=30, drop_range=(0, 30)), 1: dict(max_tokens=60, drop_range=(30, 60)),
        2: dict(max_tokens=100, drop_range=(60, 100000))}
inp = sm.SSTInputLayerV2(drop, (8, 8, 8), (80, 80, 64), shuffle_voxels=False, debug=False, mute=True).to(dev)
model = sm.SSTv2(d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2, dropout=0.0, activation='gelu',
                 num_attached_conv=0, to_bev=False, debug=False, layer_cfg=dict(compute_dtype=torch.bfloat16)).to(dev).train()
xyz, feats, bidx = synthetic_object_grids(G, P, seed=0, device=dev)
xyz[:, 2] *= 0.8
zyx = voxelization(xyz, [0.1, 0.1, 0.1], rng, -1, -1)
coors = torch.cat([bidx.view(-1, 1).to(torch.int32), zyx], 1)
vfeats, vcoors = dynamic_scatter(feats, coors, 'mean', grid_shape=[G] + list(shape))
x = torch.randn(vfeats.size(0), 128, device=dev, requires_grad=True)


def step():
    info = inp(x, vcoors.long())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = model(info)[0]['voxel_feats']
    out.float().sum().backward()
    torch.cuda.synchronize()
    return t1


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
t1 = step()
t2 = time.perf_counter()
print(f'input layer {1e3 * (t1 - t0):.2f} ms, backbone fwd + bwd {1e3 * (t2 - t1):.2f} ms (each followed by a synchronize)')
sites = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def show(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' not in str(message):
        return
    for fr in reversed(traceback.extract_stack()):
        if fr.filename.startswith(root) and 'find_syncs' not in fr.filename:
            sites[f'{os.path.relpath(fr.filename, root)}:{fr.lineno} {fr.line}'] += 1
            return
    sites[f'{filename}:{lineno}'] += 1


warnings.showwarning = show
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
info = inp(x, vcoors.long())
out = model(info)[0]['voxel_feats']
out.float().sum().backward()
torch.cuda.set_sync_debug_mode('default')
for k, v in sites.most_common():
    print(v, k)
print('total', sum(sites.values()))
