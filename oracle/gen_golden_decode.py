"""Generate tests/golden/occ_decode.npz: the REFERENCE's dense-grid decode (OccDecoder.get_occ /
get_roi_occ, mmdet3d/models/occ/occ_base.py:155-342, imported through oracle/ref_shim.py in the build
container only) on seeded RoIs with name-hashed synthetic weights.  Data only, no reference source."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_shim as R  # noqa: E402
from oracle import synth  # noqa: E402

VOXEL, SCALE, OFFSET = 0.2, [1.0, 1.0, 1.0], [0.5, 0.5, 0.5]


def main():
    R.install()
    occ_base = R.load('mmdet3d.models.occ.occ_base')
    torch.manual_seed(0)
    dec = occ_base.OccDecoder(roi_feature_channels=256, occ_mlp=[64, 128, 128], use_positional_encoding=True,
                              pos_encode_L=10, norm_pos=True, norm_cfg=dict(type='LN', eps=1e-3), act='gelu',
                              occ_dropout=0.0, cls_dim=1, pos_thresh=0.5, use_ln=True).eval()
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in dec.state_dict().items()}, seed=11)
    dec.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    R_ = 7
    rois = torch.zeros(R_, 8)
    rois[:, 0] = torch.tensor([0, 0, 0, 1, 1, 2, 2])
    rois[:, 1:4] = torch.randn(R_, 3, generator=g) * torch.tensor([20., 20., 1.])
    rois[:, 4:7] = torch.rand(R_, 3, generator=g) * torch.tensor([1.0, 3.0, 0.8]) + torch.tensor([1.6, 3.5, 1.3])
    rois[:, 7] = (torch.rand(R_, generator=g) * 2 - 1) * 3.1
    feats = torch.randn(R_, 256, generator=g)
    with torch.no_grad():
        centers = occ_base.occ_ops.generate_dense_voxel_centers(rois[:, 4:7], VOXEL, SCALE, OFFSET)
        logits = [dec.occ_forward(feats[j:j + 1].repeat(len(c), 1), c) for j, c in enumerate(centers)]
        occ = dec.get_occ(feats, rois, VOXEL, SCALE, OFFSET, transform=True)
        occ_local = dec.get_occ(feats, rois, VOXEL, SCALE, OFFSET, transform=False)
        full = dec.get_occ(feats, rois, VOXEL, SCALE, OFFSET, return_full=True, transform=True, concat_batch=True)
        pts, inds, score = dec.get_roi_occ(feats, rois, VOXEL, SCALE, OFFSET, transform=True, return_score=True,
                                           random_sample_size=0, occ_only=False)
        pts_o, inds_o = dec.get_roi_occ(feats, rois, VOXEL, SCALE, OFFSET, transform=True, occ_only=True)
    flat = [t for sample in occ for t in sample]
    flat_local = [t for sample in occ_local for t in sample]
    out = dict(
        rois=rois.numpy(), feats=feats.numpy(),
        param_names=np.array(list(sd.keys())), param_shapes=np.array([','.join(map(str, v.shape)) for v in sd.values()]),
        cells_per_roi=np.array([len(c) for c in centers]), centers=torch.cat(centers).numpy(),
        logits=torch.cat(logits).numpy().reshape(-1),
        occ_counts=np.array([len(t) for t in flat]), occ_pts=torch.cat(flat).numpy(),
        occ_local_pts=torch.cat(flat_local).numpy(),
        samples=np.array([len(s) for s in occ]),
        full_counts=np.array([len(t) for t in full]), full_pts=torch.cat(full).numpy(),
        roi_occ_pts=pts.numpy(), roi_occ_inds=inds.numpy(), roi_occ_score=score.numpy().reshape(-1),
        roi_occ_only_pts=pts_o.numpy(), roi_occ_only_inds=inds_o.numpy())
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'occ_decode.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: getattr(v, 'shape', None) for k, v in out.items()}, 'occupied', int(sum(out['occ_counts'])),
          'of', int(out['cells_per_roi'].sum()))


if __name__ == '__main__':
    main()
