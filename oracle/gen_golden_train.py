"""Generate tests/golden/ococc_train.npz by running the REFERENCE's own TrackletRoIHeadOCC (built from the
verbatim configs/ococc/ococcnet.py through oracle/ref_train_shim.py, build container only) on the seeded
scene of oracle/synth.synth_training_scene with name-hashed weights:

  * _select_one2one_candidates + TrackletAssigner + _assign_and_sample   (tracklet_roi_head_occ.py:880-1030,
    tracklet_assigner.py:14-57, lidar_tracklet.py:278-339) -> rois, frame indices, IoUs, scores, pos/neg split
  * forward_train -> the full loss dict (ococc_bbox_head.py:433-811, tracklet_roi_head_occ.py:759-826), eval mode
    (dropout off), and the gradient of  loss_rcnn_cls + loss_rcnn_bbox + loss_rcnn_occ  w.r.t. every parameter
    (norms of all 269, a few small tensors in full)
  * simple_test -> refined tracklets, and test_occ inter / union integers (:394-610)

Pooling and the BEV overlap inside come from OUR oracle (TorchEx is absent everywhere, see ref_train_shim).
The .npz holds outputs only; the inputs are regenerated from the seed by oracle/synth.py on both sides.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_train_shim as T  # noqa: E402
from oracle import synth  # noqa: E402

FULL_GRADS = ['block_list.0.rel_mlp.0.0.weight', 'block_list.5.vfe_layers.1.norm.weight',
              'occ_ae_head.point_encoder.block_list.0.vfe_layers.0.linear.weight',
              'occ_ae_head.occ_decoder.conv_occ.3.weight', 'occ_ae_head.occ_decoder.conv_occ.0.1.bias',
              'occ_ae_head.occ_decoder.ln.weight', 'trans_enc.layers.2.norm2.bias', 'roi_pos_enc_mlp.0.0.weight',
              'conv_reg.2.weight', 'conv_cls.2.weight', 'conv_fused.2.bias']


def to_ref_tracklets(c, samples):
    Trk, Boxes = c['Tracklet'], c['Boxes']

    def make(boxes, ts, scores, name):
        t = Trk('seg', name, 1, False, box_list=[Boxes(torch.from_numpy(boxes[i:i + 1].copy())) for i in range(len(boxes))],
                ts_list=list(ts), score_list=[float(s) for s in scores])
        t.set_type_name()          # 'Car'
        t.set_type(0, 'mmdet3d')   # what WaymoTrackletDataset does before the model sees it
        t.freeze()
        return t
    trks, cands, occs, occ_scores = [], [], [], []
    for b, s in enumerate(samples):
        trks.append(make(s['boxes'], s['ts'], s['scores'], f'pd{b}'))
        cands.append([make(cb, cts, np.ones(len(cb)), f'gt{b}_{j}') for j, (cb, cts, _, _) in enumerate(s['candidates'])])
        occs.append([torch.from_numpy(o) for (_, _, o, _) in s['candidates']])
        occ_scores.append([torch.tensor([sc], dtype=torch.float32) for (_, _, _, sc) in s['candidates']])
    return trks, cands, occs, occ_scores


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    head, c = T.build_reference_roi_head()
    head.eval()
    samples = synth.synth_training_scene(seed=0)
    out = {}

    # ---------------------------------------------------------------- training path
    trks, cands, occs, occ_scores = to_ref_tracklets(c, samples)
    pts = [torch.from_numpy(s['points']) for s in samples]
    cat = torch.cat(pts, 0)
    batch_idx = torch.cat([torch.full((len(p),), i, dtype=torch.long) for i, p in enumerate(pts)])
    frame_inds = torch.cat([torch.from_numpy(s['pts_frame_inds']) for s in samples])

    torch.manual_seed(123)  # the random frame shift of _assign_and_sample draws from the global generator
    fi = frame_inds.clone()
    res = head._assign_and_sample(trks, cands, occs, occ_scores, batch_idx, fi)
    for b, r in enumerate(res):
        out[f'assign_pos_inds_{b}'], out[f'assign_neg_inds_{b}'] = r.pos_inds.numpy(), r.neg_inds.numpy()
        out[f'assign_bboxes_{b}'] = r.bboxes.numpy()
        out[f'assign_frame_inds_{b}'] = r.bboxes_frame_inds.numpy()
        out[f'assign_iou_{b}'], out[f'assign_scores_{b}'] = r.iou.numpy(), r.scores.numpy()
        out[f'assign_pos_gt_bboxes_{b}'] = r.pos_gt_bboxes.numpy()
        out[f'assign_pos_gt_labels_{b}'] = r.pos_gt_labels.numpy()
    out['assign_pts_frame_inds'] = fi.numpy()

    torch.manual_seed(123)
    trks, cands, occs, occ_scores = to_ref_tracklets(c, samples)
    losses = head.forward_train(cat[:, :3], cat[:, 3:], batch_idx, frame_inds.clone(), None, trks, cands, occs, occ_scores)
    for k, v in losses.items():
        out['loss_' + k] = v.detach().numpy().astype(np.float32)
        print(f'{k:24s} {v.detach().flatten()[:4].tolist()}')
    total = losses["loss_rcnn_cls"] + losses["loss_rcnn_bbox"] + losses["loss_rcnn_occ"].mean()  # mmdet _parse_losses: mean of each entry
    head.zero_grad()
    total.backward()
    names, norms = [], []
    for n, p in head.bbox_head.named_parameters():
        names.append(n)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
    out['grad_names'], out['grad_norms'] = np.array(names), np.asarray(norms, np.float64)
    pd = dict(head.bbox_head.named_parameters())
    for n in FULL_GRADS:
        out['grad_' + n] = pd[n].grad.numpy()

    # ---------------------------------------------------------------- inference path, one tracklet at a time
    for b in range(3):
        trks, cands, occs, occ_scores = to_ref_tracklets(c, samples)
        t = trks[b]
        eye = torch.eye(4)
        t.pose_list = [eye.clone() for _ in range(len(t))]  # ego poses; identity: boxes already share one frame
        t.shared_pose = eye.clone()
        metas = [dict(box_type_3d=c['Boxes'])]
        with torch.no_grad():
            r = head.simple_test(pts[b][:, :3], pts[b][:, 3:], torch.zeros(len(pts[b]), dtype=torch.long),
                                 torch.from_numpy(samples[b]['pts_frame_inds']), metas, [t], [cands[b]], [occs[b]],
                                 [occ_scores[b]])[0]
        ot = r['out_tracklets'][0]
        out[f'test_boxes_{b}'] = np.concatenate(ot.box_list, 0).astype(np.float32)
        out[f'test_scores_{b}'] = np.asarray(ot.score_list, np.float32)
        out[f'test_inters_{b}'] = torch.cat(r['inters']).numpy() if len(r['inters']) else np.zeros(0, np.int64)
        out[f'test_unions_{b}'] = torch.cat(r['unions']).numpy() if len(r['unions']) else np.zeros(0, np.int64)
        out[f'test_gt_boxes_{b}'] = torch.cat(r['gt_boxes']).numpy() if len(r['gt_boxes']) else np.zeros((0, 7), np.float32)
        print(b, 'inters', out[f'test_inters_{b}'][:6], 'unions', out[f'test_unions_{b}'][:6])

    dst = os.path.join(HERE, '..', 'tests', 'golden', 'ococc_train.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
