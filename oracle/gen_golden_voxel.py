"""Generate tests/golden/voxelize.npz from the REFERENCE's own CPU voxeliser
(oracle/_ref/voxel_ref.so, compiled by oracle/Makefile from
/root/reference/mmdet3d/ops/voxel/src/{voxelization.cpp,voxelization_cpu.cpp,
scatter_points_cpu.cpp}).  Runs only in the build container; the .npz (inputs +
expected outputs) is committed, the reference binary is not.
"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load_ref():
    spec = importlib.util.spec_from_file_location('voxel_ref', os.path.join(HERE, '_ref', 'voxel_ref.so'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_ref()
    g = torch.Generator().manual_seed(0)
    out = {}
    cases = {
        # object grid of the benchmark: 0.2 m voxels, 40^3 cells, points spill over the box
        'grid40': dict(n=2000, nf=5, vs=[0.2, 0.2, 0.2], rng=[-4, -4, -4, 4, 4, 4], spread=4.6),
        # PosEncode box of the occupancy decoder, non cubic, voxel size not dividing the range
        'box': dict(n=1500, nf=4, vs=[0.3, 0.25, 0.2], rng=[-8, -8, -4, 8, 8, 4], spread=9.0),
        # KITTI-like cloud of the reference's own test_voxelize.py (fixture missing there)
        'kitti': dict(n=3000, nf=4, vs=[0.5, 0.5, 0.5], rng=[0, -40, -3, 70.4, 40, 1], spread=None),
    }
    for name, c in cases.items():
        if c['spread'] is None:
            pts = torch.rand(c['n'], c['nf'], generator=g)
            pts[:, 0] = pts[:, 0] * 80 - 4
            pts[:, 1] = pts[:, 1] * 90 - 45
            pts[:, 2] = pts[:, 2] * 5 - 3.5
        else:
            pts = (torch.rand(c['n'], c['nf'], generator=g) * 2 - 1) * c['spread']
        # exact cell-boundary and duplicate points exercise floor / first-come rules
        pts[:16, :3] = torch.round(pts[:16, :3] / 0.5) * 0.5
        pts[16:32] = pts[:16]
        pts = pts.contiguous()
        coors = pts.new_zeros((c['n'], 3), dtype=torch.int)
        ref.dynamic_voxelize(pts, coors, c['vs'], [float(v) for v in c['rng']], 3)
        out[f'{name}_points'] = pts.numpy()
        out[f'{name}_voxel_size'] = np.asarray(c['vs'], np.float32)
        out[f'{name}_range'] = np.asarray(c['rng'], np.float32)
        out[f'{name}_dyn_coors'] = coors.numpy()
        for mp, mv in ((5, 300), (35, 20000), (1, 7)):
            voxels = pts.new_zeros((mv, mp, c['nf']))
            hc = pts.new_zeros((mv, 3), dtype=torch.int)
            npv = pts.new_zeros((mv,), dtype=torch.int)
            vn = ref.hard_voxelize(pts, voxels, hc, npv, c['vs'], [float(v) for v in c['rng']], mp, mv, 3)
            key = f'{name}_hard_{mp}_{mv}'
            out[key + '_voxels'] = voxels[:vn].numpy()
            out[key + '_coors'] = hc[:vn].numpy()
            out[key + '_npv'] = npv[:vn].numpy()
    dst = os.path.join(HERE, '..', 'tests', 'golden', 'voxelize.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
