"""Generate tests/golden/ococc_head.npz by running the REFERENCE's own OccBBoxHead /
OccAutoEncoder / OccDecoder / SIR code (imported through oracle/ref_shim.py, build container
only) on seeded synthetic tracklets with name-hashed synthetic weights (oracle/synth.py).
The .npz holds inputs, outputs and the parameter names/shapes -- data, no reference source.

The point-pooling inputs (local_xyz, boundary_offset, is_in_margin) come from our oracle's
point_pool, because the reference's pooling kernel (TorchEx) is not available anywhere.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402
from oracle import ref_shim as R  # noqa: E402
from oracle import synth  # noqa: E402
from objectcentricocccompletion_amd import config  # noqa: E402  (loader only; no device code)


class _AttrDict(dict):
    __getattr__ = dict.get


def build_reference_head():
    mods = R.load_ococc_classes()
    cfg = config.fromfile(os.path.join(R.REF, 'configs', 'ococc', 'ococcnet.py'))
    hc = dict(cfg['model']['roi_head']['bbox_head'])
    hc.pop('type')
    head = mods['head'].OccBBoxHead(**hc)
    head.train_cfg = _AttrDict(cfg['model']['train_cfg'])
    head.test_cfg = _AttrDict(cfg['model']['test_cfg'])
    head.occ_ae_head.train_cfg = head.train_cfg
    head.occ_ae_head.test_cfg = head.test_cfg
    return head, mods


def pooled_inputs(seed=0, num_tracklets=2, frames=32, pts_per_frame=40, first_frame=5):
    t = synth.synth_tracklets(num_tracklets, frames, pts_per_frame, seed, first_frame)
    rois = t['rois']
    max_frames = int(t['roi_frame_inds'].max()) + 1
    roi_key = (rois[:, 0].astype(np.int64) * max_frames + t['roi_frame_inds']).astype(np.int32)
    pts_key = (t['pts_batch'] * max_frames + t['pts_frame']).astype(np.int32)
    pidx, ridx, feats, counts = O.point_pool(rois[:, 1:], roi_key, t['pts_xyz'], pts_key, [0.5, 0.5, 0.5], 4096, 300000)
    # decorated per-point features as PointDecoration builds them (tracklet_pipelines.py:582-607):
    # intensity, elongation, yaw/pi, size/10 (3), score ; + the RoI score appended by _bbox_forward
    r = rois[ridx]
    rng = np.random.default_rng(seed + 1)
    score = rng.uniform(0.3, 1.0, size=(len(rois),)).astype(np.float32)
    pts_feats = np.concatenate([t['pts_attr'][pidx], (r[:, 7:8] / np.pi), r[:, 4:7] / 10.0, score[ridx][:, None],
                                score[ridx][:, None]], 1).astype(np.float32)
    return dict(rois=rois, roi_frame_inds=t['roi_frame_inds'], pts_xyz=feats[:, :3].copy(), pts_feats=pts_feats,
                local_xyz=feats[:, 3:6].copy(), boundary_offset=feats[:, 6:12].copy(), is_in_margin=feats[:, 12].copy(),
                roi_inds=ridx, pool_pts_idx=pidx, roi_counts=counts, raw=t, roi_key=roi_key, pts_key=pts_key)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    head, mods = build_reference_head()
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    head.load_state_dict(synth.synth_state_dict(shapes, seed=0))
    head.eval()
    inp = pooled_inputs()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    rois, frames = T(inp['rois']), T(inp['roi_frame_inds'])
    pts_info = dict(local_xyz=T(inp['local_xyz']), boundary_offset=T(inp['boundary_offset']),
                    is_in_margin=T(inp['is_in_margin']))
    roi_inds = T(inp['roi_inds'])
    out = {}
    with torch.no_grad():
        res = head(T(inp['pts_xyz']), T(inp['pts_feats']), pts_info, roi_inds, rois, frames)
        fcf, nonempty, out_coors = head.roi_encode(T(inp['pts_xyz']), T(inp['pts_feats']), pts_info, roi_inds, rois)
        # decoder on K query points per RoI (reference-shaped call with repeated features)
        K = 48
        g = torch.Generator().manual_seed(3)
        qxyz = (torch.rand(len(rois), K, 3, generator=g) - 0.5) * torch.tensor([6.0, 3.0, 2.5])
        logits = head.occ_ae_head.occ_decoder.occ_forward(res['fused_roi_feats'][:, None, :].repeat(1, K, 1), qxyz)
        # one SIRLayer and the quantiser in isolation
        blk = head.block_list[1]
        x_in = torch.randn(len(roi_inds), 144, generator=g)
        f_cluster = torch.randn(len(roi_inds), 13, generator=g)
        pf, vf = blk(x_in, roi_inds, f_cluster)
        centers = mods['occ_ops'].quantize_points(pts_info['local_xyz'], rois, roi_inds, 0.2, to_center=True)
        pe = head.occ_ae_head.occ_decoder.pos_encode(qxyz[:4])
        tpe = head.pos_enc(frames.view(2, -1).t().float())
    # tracklets of unequal length through transformer_forward_various_length (ococc_bbox_head.py:911-995):
    # tracklet 0 keeps its 32 frames, tracklet 1 keeps 20, rows in random order
    g2 = torch.Generator().manual_seed(17)
    keep = torch.cat([torch.arange(32), 32 + torch.randperm(32, generator=g2)[:20]])
    vl_index = keep[torch.randperm(len(keep), generator=g2)]
    with torch.no_grad():
        vl = head.transformer_forward_various_length(rois[vl_index], frames[vl_index], fcf[vl_index],
                                                     nonempty[vl_index])
    out['vl_index'], out['vl_out'] = vl_index.numpy(), vl.numpy()

    # backward of one SIRLayer (voxel_encoder.py:764-832: both segment-max argmax routes, the gather-back of the
    # voxel feature, three LN+GELU pairs) and of the decoder (occ_base.py:120-139), SURVEY G1 / G4: gradients of
    # fixed random projections of the outputs w.r.t. inputs and parameters
    xg, fg = x_in.clone().requires_grad_(True), f_cluster.clone().requires_grad_(True)
    pfg, vfg = blk(xg, roi_inds, fg)
    wp, wv = torch.randn(pfg.shape, generator=g), torch.randn(vfg.shape, generator=g)
    blk.zero_grad()
    ((pfg * wp).sum() + (vfg * wv).sum()).backward()
    out['sir_wp'], out['sir_wv'] = wp.numpy(), wv.numpy()
    out['sir_grad_x'], out['sir_grad_fcluster'] = xg.grad.numpy(), fg.grad.numpy()
    for n, p_ in blk.named_parameters():
        out['sir_grad__' + n] = p_.grad.numpy().copy()
    blk.zero_grad()
    dec = head.occ_ae_head.occ_decoder
    fr = res['fused_roi_feats'].clone().requires_grad_(True)
    lg = dec.occ_forward(fr[:, None, :].repeat(1, K, 1), qxyz)
    wl = torch.randn(lg.shape, generator=g)
    dec.zero_grad()
    (lg * wl).sum().backward()
    out['dec_wl'], out['dec_grad_feats'] = wl.numpy(), fr.grad.numpy()
    for n, p_ in dec.named_parameters():   # the two 4 MB matrices as norm + leading rows only
        gq = p_.grad.numpy()
        out['dec_gradnorm__' + n] = np.float64(np.linalg.norm(gq.astype(np.float64)))
        out['dec_grad__' + n] = gq.copy() if gq.size <= 70000 else gq[:32].copy()
    dec.zero_grad()

    for k in ('fused_roi_feats', 'ori_roi_feats', 'cls_score', 'bbox_pred'):
        out['out_' + k] = res[k].numpy()
    out['out_nonempty_roi_mask'] = res['nonempty_roi_mask'].numpy()
    out['out_final_cluster_feats'] = fcf.numpy()
    out['dec_xyz'], out['dec_logits'] = qxyz.numpy(), logits.numpy()
    out['sir_x'], out['sir_fcluster'] = x_in.numpy(), f_cluster.numpy()
    out['sir_point_feats'], out['sir_voxel_feats'] = pf.numpy(), vf.numpy()
    out['quant_centers'] = centers.numpy()
    out['posenc_in'], out['posenc_out'] = qxyz[:4].numpy(), pe.numpy()
    out['tpe_out'] = tpe.numpy()

    # targets (pure reference tensor code) on a synthetic assignment: GT = RoI + noise
    rng = np.random.default_rng(11)
    B, L = 2, 32
    samples, occ_all = [], []
    for b in range(B):
        rb = inp['rois'][inp['rois'][:, 0] == b][:, 1:]
        gt = rb + rng.normal(0, [0.1, 0.1, 0.1, 0.05, 0.05, 0.05, 0.02], size=rb.shape).astype(np.float32)
        iou = rng.uniform(0.1, 1.0, size=(L,)).astype(np.float32)
        occ = np.concatenate([(rng.random((40, 3)) - 0.5) * [5, 2.2, 1.8], rng.integers(0, 3, size=(40, 1))], 1).astype(np.float32)
        s = _AttrDict(pos_bboxes=T(rb), pos_gt_bboxes=T(gt.astype(np.float32)), iou=T(iou),
                      pos_gt_labels=torch.zeros(L, dtype=torch.long), occ_labels=T(occ), occ_scores=T(np.array([0.9 - 0.6 * b], np.float32)))
        samples.append(s)
        out[f'tgt_in_gt_{b}'], out[f'tgt_in_iou_{b}'], out[f'tgt_in_occ_{b}'] = gt.astype(np.float32), iou, occ
    tg = head.get_targets(samples, head.train_cfg, transform_occ=True, num_occ_per_tracklet=-1)
    names = ['label', 'bbox_targets', 'bbox_target_batch_idx', 'pos_gt_bboxes', 'pos_gt_labels', 'reg_mask',
             'label_weights', 'bbox_weights', 'pos_roi_local_xyz', 'gt_occ', 'occ_score', 'occ_reg_mask',
             'occ_target_batch_idx', 'pos_gt_bboxes_occ']
    for n, v in zip(names, tg):
        out['tgt_' + n] = v.numpy()
    dec = head.decode_from_rois(rois, res['bbox_pred'])
    out['decoded_boxes'] = dec.detach().numpy()

    for k in ('rois', 'roi_frame_inds', 'pts_xyz', 'pts_feats', 'local_xyz', 'boundary_offset', 'is_in_margin', 'roi_inds'):
        out['in_' + k] = inp[k]
    out['param_names'] = np.array(list(shapes.keys()))
    out['param_shapes'] = np.array([','.join(map(str, s)) for s in shapes.values()])
    dst = os.path.join(HERE, '..', 'tests', 'golden', 'ococc_head.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes;', len(shapes), 'params;', len(roi_inds), 'pooled points;',
          int(res['nonempty_roi_mask'].sum()), 'non-empty rois of', len(rois))


if __name__ == '__main__':
    main()
