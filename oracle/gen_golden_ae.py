"""Generate tests/golden/occ_ae.npz: the REFERENCE's auto-encoder stage (OccAutoEncoder.sample_observation,
forward_train_ae's deterministic path, loss, online_tuning_forward; occ_ae_head.py:65-201,270-391,451-509) run
through oracle/ref_shim.py in the build container on the same seeded inputs / synthetic weights as
ococc_head.npz.  Data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import synth  # noqa: E402
from oracle.gen_golden_ococc import build_reference_head, pooled_inputs  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    head, _ = build_reference_head()
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    head.load_state_dict(synth.synth_state_dict(shapes, seed=0))
    ae = head.occ_ae_head.eval()

    # mmdet's CrossEntropyLoss(use_sigmoid=True, reduction='none') is external to the reference checkout (the
    # shim builds None for it): its documented arithmetic, element-wise BCE-with-logits times the weight
    def bce(pred, label, weight=None, **kw):
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, label.float(), reduction='none')
        return loss if weight is None else loss * weight.float()
    ae.loss_occ_ae = bce
    inp = pooled_inputs()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    rois, roi_inds = T(inp['rois']), T(inp['roi_inds'])
    pts_info = dict(local_xyz=T(inp['local_xyz']), boundary_offset=T(inp['boundary_offset']),
                    is_in_margin=T(inp['is_in_margin']))
    out = {}
    with torch.no_grad():
        feats, nonempty, local_xyz = ae.encode(T(inp['pts_xyz']), T(inp['pts_feats'])[:, :2], pts_info, roi_inds, rois)
        xyz, labels, inds = ae.sample_observation(local_xyz, rois, roi_inds, downsample_size=-1, balance_sample=False)
        preds = ae.decode(feats, xyz, inds)
        loss = ae.loss(preds, feats, xyz, inds, labels, nonempty)
    out['enc_feats'], out['enc_nonempty'], out['enc_local_xyz'] = feats.numpy(), nonempty.numpy(), local_xyz.numpy()
    out['obs_labels'], out['obs_inds'] = labels.numpy().astype(np.int8), inds.numpy().astype(np.int32)
    out['obs_xyz_head'] = xyz[:4096].numpy()
    out['obs_count_per_roi'] = np.bincount(inds.numpy(), minlength=len(rois))
    out['obs_pos_per_roi'] = np.bincount(inds.numpy(), weights=labels.numpy(), minlength=len(rois)).astype(np.int64)
    out['dec_preds_head'] = preds[:4096].numpy().reshape(-1)
    for k, v in loss.items():
        out['loss_' + k] = np.asarray(float(v))
    # test-time tuning on the first 6 RoIs' cells (3 Adam steps)
    sel = inds < 6
    tuned = ae.online_tuning_forward(feats[:6], xyz[sel], labels[sel], None, inds[sel], 3)
    out['tuned'] = tuned.detach().numpy()
    dst = os.path.join(HERE, '..', 'tests', 'golden', 'occ_ae.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes; cells', len(labels), 'observed', int(labels.sum()),
          {k: float(v) for k, v in loss.items()})


if __name__ == '__main__':
    main()
