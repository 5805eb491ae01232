import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from objectcentricocccompletion_amd.graph import GraphedStep
from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = SubMOccEncoder(grouped_points=True).to(dev)
B = 4
xyz, feats, bidx = synthetic_object_grids(B, 500, seed=3, device=dev)
d = torch.zeros(xyz.shape[0], 128, dtype=torch.bfloat16, device=dev)
mode = sys.argv[1]
def fwd_bwd():
    model.zero_grad(set_to_none=True)
    out = model(xyz, feats, bidx, B, static=True)
    if mode != 'fwd':
        out.features.backward(d)
    return out
with torch.set_grad_enabled(mode != 'fwd'):
    g = GraphedStep(fwd_bwd, warmup=2)
    g.replay(); torch.cuda.synchronize(); print('replay1 ok', flush=True)
    xyz2 = xyz.roll(17, 0).contiguous()
    if 'eager' in mode:
        want = model(xyz2, feats, bidx, B, static=True).features.clone()
        torch.cuda.synchronize(); print('eager xyz2 ok', flush=True)
    if 'dyn' in mode:
        want = model(xyz2, feats, bidx, B).features.clone()
        torch.cuda.synchronize(); print('dyn xyz2 ok', flush=True)
    if 'alloc' in mode:
        zz = [torch.randn(1 << 20, device=dev) for _ in range(8)]
        torch.cuda.synchronize(); print('alloc ok', flush=True)
    if 'same' in mode:
        want = model(xyz, feats, bidx, B, static=True).features.clone()
        torch.cuda.synchronize(); print('eager same ok', flush=True)
    if 'skipcopy' in mode:
        xyz2 = xyz.clone()
    xyz.copy_(xyz2)
    torch.cuda.synchronize(); print('copy ok', flush=True)
    o = g.replay(); torch.cuda.synchronize(); print('replay2 ok', flush=True)
