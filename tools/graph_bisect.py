"""Capture single stages of the encoder step in a HIP graph, one subprocess per stage, to find
which launch sequence the graph runtime rejects.  Usage: python tools/graph_bisect.py [stage]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGES = ['seg_fill', 'seg_sum', 'seg_mean', 'seg_sum_c4', 'cast', 'scatter_nocast', 'torch_only', 'memset_only', 'voxelize', 'unique', 'scatter', 'rulebook', 'conv_fwd', 'ln_fwd', 'fwd', 'fwd_bwd', 'adamw']


def run(stage):
    import torch
    from objectcentricocccompletion_amd.graph import GraphedStep
    from objectcentricocccompletion_amd.occ_encoder import SubMOccEncoder, synthetic_object_grids
    from objectcentricocccompletion_amd.voxel import dynamic_scatter
    from objectcentricocccompletion_amd.voxel.scatter_points import grid_unique, segment_reduce
    from objectcentricocccompletion_amd.spconv import SparseConvTensor
    from objectcentricocccompletion_amd.spconv import ops
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model = SubMOccEncoder(grouped_points=True).to(dev)
    B = 4
    xyz, feats, bidx = synthetic_object_grids(B, 500, seed=3, device=dev)
    coors = model.voxelize(xyz, bidx, B)
    dims = [B] + model.sparse_shape
    vf, vc = dynamic_scatter(feats, coors, 'mean', grid_shape=dims, static=True)
    x0 = SparseConvTensor(vf.to(torch.bfloat16), vc, model.sparse_shape, B)
    d = torch.zeros(xyz.shape[0], 128, dtype=torch.bfloat16, device=dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        out = model(xyz, feats, bidx, B, static=True)
        out.features.backward(d)
        return out

    if stage == 'adamw':
        fwd_bwd()
    buf = torch.zeros(1 << 16, device=dev)
    from objectcentricocccompletion_amd import _lib as L
    u_c, u_inv, u_cnt, _ = grid_unique(coors, dims, static=True)
    seg_out = torch.empty(u_c.shape[0], feats.shape[1], device=dev)
    def seg_fill():
        L.check(L.lib.ococc_segment_reduce_f32(L.ptr(feats), L.ptr(u_inv), 0, feats.shape[1], 1, L.ptr(u_cnt), L.ptr(seg_out), None, u_c.shape[0], L.stream()))
        return seg_out
    f4 = feats[:, :4].contiguous()
    fns = {
        'seg_fill': seg_fill,
        'seg_sum': lambda: segment_reduce(feats, u_inv, u_c.shape[0], 'sum', u_cnt),
        'seg_mean': lambda: segment_reduce(feats, u_inv, u_c.shape[0], 'mean', u_cnt),
        'seg_sum_c4': lambda: segment_reduce(f4, u_inv, u_c.shape[0], 'sum', u_cnt),
        'cast': lambda: vf.to(torch.bfloat16),
        'scatter_nocast': lambda: dynamic_scatter(feats, coors, 'sum', grid_shape=dims, static=True),
        'torch_only': lambda: torch.cat([xyz, xyz], 1) * 2,
        'memset_only': lambda: buf.zero_(),
        'voxelize': lambda: model.voxelize(xyz, bidx, B),
        'unique': lambda: grid_unique(coors, dims, static=True),
        'scatter': lambda: dynamic_scatter(feats, coors, 'mean', grid_shape=dims, static=True),
        'rulebook': lambda: ops.get_indice_pairs(vc, B, model.sparse_shape, 3, 1, 1, 1, 0, True),
        'conv_fwd': lambda: model.conv_layers[0][0](x0),
        'ln_fwd': lambda: model.conv_layers[0](x0),
        'fwd': lambda: model(xyz, feats, bidx, B, static=True),
        'fwd_bwd': fwd_bwd,
        'adamw': opt.step,
    }
    with torch.set_grad_enabled(stage in ('fwd_bwd', 'adamw')):
        g = GraphedStep(fns[stage], warmup=2 if stage != 'adamw' else 0)
        g.replay()
        torch.cuda.synchronize()
        zz = [torch.randn(1 << 20, device=dev) for _ in range(8)]  # fresh device allocations between replays
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
    print('OK', stage, flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for s in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True)
            tail = (r.stdout + r.stderr).strip().splitlines()
            msg = [l for l in tail if 'Error' in l or 'error' in l or 'Fatal' in l][:2]
            print(f'{s:10s} rc={r.returncode} {"OK" if "OK " + s in r.stdout else "FAIL"} {msg}', flush=True)
