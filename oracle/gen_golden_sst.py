"""Generate tests/golden/sst.npz from the REFERENCE's SST code (SSTInputLayerV2, SSTv2,
BasicShiftBlockV2, WindowAttention and the window helpers of sst_ops.py), imported through
oracle/ref_shim.py in the build container.  get_inner_win_inds is the reference's own pure-torch
twin (get_inner_win_inds_deprecated, sst_ops.py:194-241) because the TorchEx kernel is absent.
Window populations stay below the largest drop level so no voxel is dropped and the result does
not depend on the (unspecified) order of voxels inside a window."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim as R  # noqa: E402
from oracle import synth  # noqa: E402

DROP = {0: dict(max_tokens=30, drop_range=(0, 30)), 1: dict(max_tokens=60, drop_range=(30, 60)),
        2: dict(max_tokens=100, drop_range=(60, 100000))}
SPARSE, WINDOW = (40, 40, 32), (8, 8, 8)   # (x, y, z); the reference asserts z < x


def main():
    R.install()
    sst_ops = sys.modules['mmdet3d.ops.sst.sst_ops']
    sst_ops.get_inner_win_inds = sst_ops.get_inner_win_inds_deprecated
    sys.modules['mmdet3d.ops'].get_inner_win_inds = sst_ops.get_inner_win_inds_deprecated
    sys.modules['mmcv.cnn'].build_conv_layer = None
    b = sys.modules['mmdet3d.models.builder']
    b.MIDDLE_ENCODERS = R.REG['VOXEL_ENCODERS']
    inp = R.load('mmdet3d.models.middle_encoders.sst_input_layer_v2')
    R.load('mmdet3d.models.sst.sst_basic_block_v2')
    bb = R.load('mmdet3d.models.backbones.sst_v2')
    g = torch.Generator().manual_seed(0)
    B = 2
    coors = []
    for b_ in range(B):  # sparse background + one dense 16^3 corner so that all three drop levels occur
        bg = torch.randperm(SPARSE[0] * SPARSE[1] * SPARSE[2], generator=g)[:400]
        bg = torch.stack([bg // (SPARSE[0] * SPARSE[1]), (bg // SPARSE[0]) % SPARSE[1], bg % SPARSE[0]], 1)
        dn = torch.randperm(16 ** 3, generator=g)[:520]
        dn = torch.stack([dn // 256 + 8 * b_, (dn // 16) % 16 + 16, dn % 16 + 8], 1)
        c = torch.unique(torch.cat([bg, dn]), dim=0)
        coors.append(torch.cat([torch.full((len(c), 1), b_), c], 1))
    coors = torch.cat(coors).long()   # (b, z, y, x)
    feats = torch.randn(len(coors), 128, generator=g)
    layer = inp.SSTInputLayerV2(DROP, WINDOW, SPARSE, shuffle_voxels=False, debug=True, mute=True).eval()
    info = layer(feats, coors)
    assert len(info['voxel_feats']) == len(feats), 'a voxel was dropped: lower the density'
    model = bb.SSTv2(d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2, dropout=0.0,
                     activation='gelu', num_attached_conv=0, to_bev=False, debug=True).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synth.synth_state_dict(shapes, seed=7))
    with torch.no_grad():
        out = model(info)[0]['voxel_feats']
        blk = model.block_list[0].encoder_list[0]
        one = blk(feats, info['pos_dict_shift0'], info['flat2win_inds_shift0'], info['key_mask_shift0'])
    # cosine attention variant of the same block (layer_cfg cosine / non_shared_tau, cosine_msa.py:123-185,449-466)
    blkmod = sys.modules['mmdet3d.models.sst.sst_basic_block_v2']
    cos = blkmod.EncoderLayer(128, 8, 256, 0.0, 'gelu', batch_first=False, layer_id=0,
                              layer_cfg=dict(cosine=True, tau_min=0.01, non_shared_tau=True)).eval()
    cshapes = {k: tuple(v.shape) for k, v in cos.state_dict().items()}
    csd = synth.synth_state_dict(cshapes, seed=9)
    csd['win_attn.self_attn.tau'] = torch.linspace(0.05, 0.4, 8).view(1, 8, 1, 1)
    cos.load_state_dict(csd)
    with torch.no_grad():
        cos_out = cos(feats, info['pos_dict_shift0'], info['flat2win_inds_shift0'], info['key_mask_shift0'])
    res = dict(coors=coors.numpy(), feats=feats.numpy(), out=out.numpy(), one_layer=one.numpy(),
               cos_out=cos_out.numpy(), cos_param_names=np.array(list(cshapes)),
               cos_param_shapes=np.array([','.join(map(str, s_)) for s_ in cshapes.values()]),
               param_names=np.array(list(shapes)), param_shapes=np.array([','.join(map(str, s)) for s in shapes.values()]))
    for i in range(2):
        res[f'batch_win_inds_shift{i}'] = info[f'batch_win_inds_shift{i}'].numpy()
        res[f'coors_in_win_shift{i}'] = info[f'coors_in_win_shift{i}'].numpy()
        res[f'drop_level_shift{i}'] = info[f'voxel_drop_level_shift{i}'].numpy()
        # order independent views of the padded layout: pos-embed / mask gathered back per voxel
        pos_flat = sst_ops.window2flat_v2(info[f'pos_dict_shift{i}'], info[f'flat2win_inds_shift{i}'])
        res[f'pos_flat_shift{i}'] = pos_flat[::4].numpy()
        res[f'tokens_per_level_shift{i}'] = np.array([int((~m).sum()) for m in info[f'key_mask_shift{i}'].values()])
    dst = os.path.join(HERE, '..', 'tests', 'golden', 'sst.npz')
    np.savez_compressed(dst, **res)
    print('wrote', dst, os.path.getsize(dst), 'bytes;', len(coors), 'voxels; levels', sorted(set(info['voxel_drop_level_shift0'].tolist())), sorted(set(info['voxel_drop_level_shift1'].tolist())), 'max window population',
          int(torch.bincount(info['batch_win_inds_shift0']).max()), int(torch.bincount(info['batch_win_inds_shift1']).max()))


if __name__ == '__main__':
    main()
