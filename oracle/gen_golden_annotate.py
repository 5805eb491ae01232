"""Generate tests/golden/occ_annotate.npz: the REFERENCE's range-image projection
point_cloud_to_range_image_idx (tools/occ/occ_annotate.py:141-207) run in the build container on seeded
inputs, and the visibility labels annotate_trk derives from it (:519-540).

The module itself cannot be imported (argparse and Waymo I/O at import time), so the function object is built
from the reference file's own syntax tree at generation time -- nothing of it is stored here; the three lines
that turn its outputs into labels (gather, `>=`, max over frames and sensors) are restated below.
Data only, no reference source."""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('OCOCC_REFERENCE', '/root/reference')


def reference_function():
    path = os.path.join(REF, 'tools', 'occ', 'occ_annotate.py')
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'point_cloud_to_range_image_idx']
    assert len(fn) == 1
    ns = {'torch': torch, 'np': np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, 'exec'), ns)
    return ns['point_cloud_to_range_image_idx']


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def main():
    f = reference_function()
    rng = np.random.default_rng(17)
    S, F, N, H, W = 2, 4, 1800, 64, 512
    # object-frame cell centres of a 4.8 x 2.0 x 1.8 m box at 0.2 m, a random subset ("unoccupied" cells)
    gx, gy, gz = np.meshgrid(np.arange(24), np.arange(10), np.arange(9), indexing='ij')
    cells = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    cells = cells[rng.permutation(len(cells))[:N]]
    centers = cells.astype(np.float64) * 0.2 + np.array([-2.4, -1.0, 0.0]) + 0.1
    N = len(centers)
    boxes = np.zeros((F, 7), np.float32)
    boxes[:, 0] = np.linspace(12, 30, F) * np.array([1, -1, 1, 1])[:F]
    boxes[:, 1] = np.linspace(-8, 15, F)
    boxes[:, 2] = rng.uniform(-0.2, 0.4, F)
    boxes[:, 3:6] = [2.0, 4.8, 1.8]
    boxes[:, 6] = rng.uniform(-3.1, 3.1, F)
    ext = np.zeros((S, F, 4, 4))
    for s in range(S):
        for k in range(F):
            R = rot_z(rng.uniform(-3.1, 3.1) if s else rng.uniform(-0.05, 0.05)) @ rot_y(rng.uniform(-0.03, 0.03))
            ext[s, k, :3, :3] = R
            ext[s, k, :3, 3] = [1.43 + rng.normal(0, 0.2), rng.normal(0, 0.3), 2.18 - 0.9 * s]
            ext[s, k, 3, 3] = 1
    inc = np.sort(rng.uniform(-0.31, 0.04, (S, F, H)), -1)[..., ::-1].copy()   # flipped beam table: descending
    # ego-frame centres per frame, exactly as annotate_trk builds them (float32 sin / cos widened to float64)
    uc = torch.from_numpy(centers)
    ego = []
    for k in range(F):
        rz = torch.tensor(boxes[k, 6])
        sn, cs = torch.sin(rz), torch.cos(rz)
        rot_t = torch.tensor([[cs, -sn, 0], [sn, cs, 0], [0, 0, 1]], dtype=uc.dtype)
        ego.append(uc @ rot_t + torch.from_numpy(boxes[k:k + 1, :3].astype(np.float64)))
    ego = torch.stack(ego, 0)
    out = dict(centers=centers, boxes=boxes, extrinsics=ext, inclinations=inc, size=np.array([H, W]))
    vis_all = []
    for s in range(S):
        idx, ri_range = f(ego, torch.from_numpy(ext[s]), torch.from_numpy(inc[s]), (H, W))
        out[f'idx_{s}'] = idx.numpy().astype(np.int32)
        out[f'range_{s}'] = ri_range.numpy()
        # range images: mostly returns behind or in front of the cell by a margin, so that the labels do not
        # hinge on the last bit of the range; a few exact ties exercise the `>=`
        img = rng.uniform(2.0, 9.0, (F, H, W)).astype(np.float32)    # background: returns in front of the object
        for k in range(F):
            r, c = idx[k, :, 0].numpy(), idx[k, :, 1].numpy()
            sel = rng.random(N) < 0.008
            delta = np.where(rng.random(N) < 0.5, 0.5, -0.5)
            img[k, r[sel], c[sel]] = (ri_range[k].numpy()[sel] + delta[sel]).astype(np.float32)
        out[f'range_image_{s}'] = img
        ri_values = torch.stack([torch.from_numpy(img[k])[idx[k, :, 0].long(), idx[k, :, 1].long()] for k in range(F)], 0)
        visibility = torch.zeros_like(ri_values, dtype=torch.int32)
        visibility[(ri_values >= ri_range)] = 2
        vis_all.append(visibility.max(0)[0])
    out['visibility'] = torch.stack(vis_all, 0).max(0)[0].numpy()
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'occ_annotate.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, 'visible fraction', float((out['visibility'] == 2).mean()))


if __name__ == '__main__':
    main()
