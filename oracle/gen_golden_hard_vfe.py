"""Generate tests/golden/hard_vfe.npz from the REFERENCE's hard-layout voxel encoders (HardSimpleVFE, HardVFE with its
VFELayers: mmdet3d/models/voxel_encoders/voxel_encoder.py:18-50, 301-500, utils.py:8-104), imported through
oracle/ref_shim.py in the build container.  Inputs: a seeded [voxels, max_points, C] block with ragged populations (zero
rows behind them, as hard_voxelize leaves them), the module's seeded parameters and BatchNorm running statistics; outputs
in eval mode and in training mode (batch statistics), plus the gradient of sum(out^2) w.r.t. the first linear weight."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim as R  # noqa: E402

CFG = dict(in_channels=5, feat_channels=[16, 32], with_distance=False, with_cluster_center=True, with_voxel_center=True,
           voxel_size=(0.5, 0.5, 1.0), point_cloud_range=(0.0, -4.0, -2.0, 8.0, 4.0, 2.0),
           norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01))


def inputs():
    g = torch.Generator().manual_seed(11)
    V, M = 37, 10
    num = torch.randint(1, M + 1, (V,), generator=g)
    num[3], num[7] = 1, M
    coors = torch.stack([torch.randint(0, 2, (V,), generator=g), torch.randint(0, 4, (V,), generator=g),
                         torch.randint(0, 16, (V,), generator=g), torch.randint(0, 16, (V,), generator=g)], 1).int()
    feats = torch.rand(V, M, 5, generator=g)
    feats[:, :, 0] = (coors[:, 3:4] + feats[:, :, 0]) * 0.5
    feats[:, :, 1] = (coors[:, 2:3] + feats[:, :, 1]) * 0.5 - 4.0
    feats[:, :, 2] = (coors[:, 1:2] + feats[:, :, 2]) * 1.0 - 2.0
    feats = feats * (torch.arange(M)[None, :] < num[:, None]).unsqueeze(-1)
    return feats, num, coors


def dynamic_inputs(feats, num, coors):
    keep = torch.arange(feats.shape[1])[None, :] < num[:, None]
    pts = feats[keep]
    pc = coors[:, None, :].expand(-1, feats.shape[1], -1)[keep]
    perm = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))
    pts, pc = pts[perm].contiguous(), pc[perm].contiguous()
    bounds = torch.tensor([0.0, -4.0, -2.0]).expand(pts.shape[0], 3) + 0.01 * pc[:, :1].float()   # (a grid origin per sample)
    return pts, pc, bounds.contiguous()


def main():
    R.install()
    ve = R.load('mmdet3d.models.voxel_encoders.voxel_encoder')
    feats, num, coors = inputs()
    out = dict(feats=feats.numpy(), num=num.numpy(), coors=coors.numpy())
    out['simple'] = ve.HardSimpleVFE(num_features=4)(feats, num, coors).numpy()
    torch.manual_seed(5)
    m = ve.HardVFE(**CFG)
    with torch.no_grad():
        for layer in m.vfe_layers:
            layer.norm.running_mean.normal_(0, 0.1)
            layer.norm.running_var.uniform_(0.5, 1.5)
            layer.norm.weight.uniform_(0.5, 1.5)
            layer.norm.bias.normal_(0, 0.1)
    for k, v in m.state_dict().items():
        out['p.' + k] = v.numpy().copy()   # (a view would follow the running statistics through the training pass below)
    m.eval()
    with torch.no_grad():
        out['eval'] = m(feats, num, coors).numpy()
    m.train()
    y = m(feats, num, coors)
    y.pow(2).sum().backward()
    out['train'] = y.detach().numpy()
    out['train_dw0'] = m.vfe_layers[0].linear.weight.grad.numpy()
    out['pad'] = ve.get_paddings_indicator(num, 10, axis=0).numpy() if hasattr(ve, 'get_paddings_indicator') else \
        sys.modules['mmdet3d.models.voxel_encoders.utils'].get_paddings_indicator(num, 10, axis=0).numpy()
    # the dynamic layout's scatter variants (:503-683), on the points of the same voxels in a shuffled order
    pts, pc, bounds = dynamic_inputs(feats, num, coors)
    out.update(d_pts=pts.numpy(), d_coors=pc.numpy(), d_bounds=bounds.numpy())
    for tag, cls, extra in (('ds', ve.DynamicScatterVFE, ()), ('dr', ve.DynamicRangeScatterVFE, (bounds,))):
        torch.manual_seed(6)
        d = cls(**dict(CFG, mode='max', rel_dist_scaler=10.0, unique_once=True)).eval()
        with torch.no_grad():
            for layer in d.vfe_layers:
                layer.norm.running_mean.normal_(0, 0.1)
                layer.norm.running_var.uniform_(0.5, 1.5)
            for k, v in d.state_dict().items():
                out[tag + '.p.' + k] = v.numpy().copy()
            vf, vc, inv = d(pts, pc, *extra, return_inv=True)
        out[tag + '.feats'], out[tag + '.coors'], out[tag + '.inv'] = vf.numpy(), vc.numpy(), inv.numpy()
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'hard_vfe.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items() if not k.startswith('p.')})


if __name__ == '__main__':
    main()
