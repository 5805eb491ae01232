"""CPU suite: the product's OcOccNet host logic (registry-built TrackletDetectorOCC: assignment, targets, losses,
test_occ, tracklet update) with its HIP leaf operators swapped for torch / oracle restatements (oracle/cpu_port.py,
test infrastructure), against the REFERENCE's outputs on the same scene (tests/golden/ococc_train.npz).  Runs without a
GPU; the same assertions run on the HIP path in tests/test_gpu_ococc_train.py."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_port, synth


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'ococc_train.npz'))


@pytest.fixture(scope='module')
def model():
    torch.set_num_threads(8)
    return cpu_port.build_detector_cpu().eval()


def _scene():
    from objectcentricocccompletion_amd.tracklet import Tracklet
    samples = synth.synth_training_scene(seed=0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    points = [T(s['points']) for s in samples]
    frames = [T(s['pts_frame_inds']) for s in samples]
    trks = [Tracklet(T(s['boxes']), s['ts'], T(s['scores']), type=0) for s in samples]
    cands = [[Tracklet(T(cb), cts, type=0) for (cb, cts, _, _) in s['candidates']] for s in samples]
    occs = [[T(o) for (_, _, o, _) in s['candidates']] for s in samples]
    occ_scores = [[torch.tensor([sc], dtype=torch.float32) for (_, _, _, sc) in s['candidates']] for s in samples]
    return samples, points, frames, trks, cands, occs, occ_scores


def test_forward_train_losses_on_cpu_port_equal_reference(gold, model):
    samples, points, frames, trks, cands, occs, occ_scores = _scene()
    torch.manual_seed(123)
    with cpu_port.cpu_ops():
        losses = model(return_loss=True, points=points, pts_frame_inds=[f.clone() for f in frames], img_metas=None,
                       tracklet=trks, gt_tracklet_candidates=cands, occ_labels=occs, occ_labels_scores=occ_scores)
    for k, v in losses.items():
        got, exp = v.detach().float().numpy().reshape(-1), gold['loss_' + k].reshape(-1)
        assert got.shape == exp.shape, k
        if k.startswith('num_'):
            assert np.array_equal(got, exp), k
        else:
            assert np.allclose(got, exp, rtol=1e-4, atol=2e-4), (k, np.abs(got - exp).max())


@pytest.mark.parametrize('b', [0, 1])
def test_simple_test_on_cpu_port_equals_reference(gold, model, b):
    samples, points, frames, trks, cands, occs, occ_scores = _scene()
    t = trks[b]
    t.pose_list, t.shared_pose = [torch.eye(4) for _ in range(len(t))], torch.eye(4)
    with cpu_port.cpu_ops(), torch.no_grad():
        r = model(return_loss=False, points=[points[b]], img_metas=[dict()], pts_frame_inds=[frames[b]], tracklet=[t],
                  gt_tracklet_candidates=[cands[b]], occ_labels=[occs[b]], occ_labels_scores=[occ_scores[b]])[0]
    ot = r['out_tracklets'][0]
    assert np.allclose(ot.boxes.numpy(), gold[f'test_boxes_{b}'], rtol=1e-4, atol=2e-4)
    assert np.array_equal(torch.cat(r['inters']).numpy(), gold[f'test_inters_{b}'])
    assert np.array_equal(torch.cat(r['unions']).numpy(), gold[f'test_unions_{b}'])


def test_batched_targets_equal_the_per_tracklet_loop(model):
    """get_targets: rows of all tracklets concatenated first (the product's path) against the reference-shaped loop over
    tracklets, on the scene with negatives, a frame without a match and a tracklet without a candidate -- bit for bit."""
    samples, points, frames, trks, cands, occs, occ_scores = _scene()
    head = model.roi_head.bbox_head
    seen = {}

    def spy(results, cfg, concat=True, transform_occ=True, num_occ_per_tracklet=-1):
        seen['batched'] = head._get_targets_batched(results, cfg, transform_occ, num_occ_per_tracklet)
        seen['loop'] = head._get_targets_per_tracklet(results, cfg, transform_occ, num_occ_per_tracklet)
        return seen['batched']

    head.get_targets = spy
    try:
        torch.manual_seed(123)
        with cpu_port.cpu_ops():
            model(return_loss=True, points=points, pts_frame_inds=[f.clone() for f in frames], img_metas=None, tracklet=trks,
                  gt_tracklet_candidates=cands, occ_labels=occs, occ_labels_scores=occ_scores)
    finally:
        del head.get_targets
    assert len(seen['batched']) == len(seen['loop']) == 14
    for i, (a, b) in enumerate(zip(seen['batched'], seen['loop'])):
        assert a.shape == b.shape and a.dtype == b.dtype, (i, a.shape, b.shape, a.dtype, b.dtype)
        assert torch.equal(a, b), i
