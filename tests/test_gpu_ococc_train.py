"""GPU parity of the training / evaluation path of TrackletRoIHeadOCC against the REFERENCE's own
TrackletRoIHeadOCC run on the same seeded scene with the same name-hashed weights
(tests/golden/ococc_train.npz, oracle/gen_golden_train.py):

  A2   _select_one2one_candidates / TrackletAssigner / _assign_and_sample
  A13  the loss dict of forward_train (every key) and the parameter gradients behind it
  A14  simple_test: refined tracklets and the test_occ inter / union integers

The scene is regenerated from its seed (oracle/synth.synth_training_scene); the .npz holds the reference's
outputs only."""
import os

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu

def test_graph_pairs_soak_in_a_fresh_process(dev):
    """5 000 training steps of the configs[2] model with the product's default switches -- the temporal transformer and
    the head's tail replayed as HIP-graph pairs, the backward replays on the calling thread (heads.GRAPH_AUTOGRAD_THREADS) --
    in a fresh child process that has to end with exit code 0: a child killed by a signal (round 3 saw ONE segmentation
    fault inside hipGraphLaunch, with the backward graph replayed from autograd's device thread) fails the test; nothing
    is retried (tools/soak_graph_pairs.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'soak_graph_pairs.py'), '--steps', '5000'],
                       capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'graph pairs in use: 2' in r.stdout and 'autograd multithreading=False' in r.stdout, r.stdout[-1500:]


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'ococc_train.npz'), allow_pickle=False)


@pytest.fixture(scope='module')
def model(dev):
    from objectcentricocccompletion_amd import heads, point_pool, roi_head  # noqa: F401 (register)
    from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg
    from objectcentricocccompletion_amd.registry import DETECTORS
    m = DETECTORS.build(ococcnet_model_cfg())
    bh = m.roi_head.bbox_head
    bh.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in bh.state_dict().items()}, seed=0))
    return m.to(dev).eval()   # eval: dropout off, as in the generator


def _scene(dev):
    from objectcentricocccompletion_amd.tracklet import Tracklet
    samples = synth.synth_training_scene(seed=0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    points = [T(s['points']) for s in samples]
    frames = [T(s['pts_frame_inds']) for s in samples]
    trks = [Tracklet(T(s['boxes']), s['ts'], T(s['scores']), type=0) for s in samples]
    cands = [[Tracklet(T(cb), cts, type=0) for (cb, cts, _, _) in s['candidates']] for s in samples]
    occs = [[T(o) for (_, _, o, _) in s['candidates']] for s in samples]
    occ_scores = [[torch.tensor([sc], dtype=torch.float32, device=dev) for (_, _, _, sc) in s['candidates']]
                  for s in samples]
    return samples, points, frames, trks, cands, occs, occ_scores


def test_assign_and_sample_equals_reference(dev, gold, model):
    samples, points, frames, trks, cands, occs, occ_scores = _scene(dev)
    batch_idx = torch.cat([torch.full((len(p),), i, dtype=torch.long, device=dev) for i, p in enumerate(points)])
    fi = torch.cat(frames).clone()
    torch.manual_seed(123)
    res = model.roi_head._assign_and_sample(trks, cands, occs, occ_scores, batch_idx, fi)
    assert len(res) == 4
    for b, r in enumerate(res):
        g = lambda k: gold[f'assign_{k}_{b}']
        assert np.array_equal(r.pos_inds.cpu().numpy(), g('pos_inds'))
        assert np.array_equal(r.neg_inds.cpu().numpy(), g('neg_inds'))
        assert np.array_equal(r.bboxes.cpu().numpy(), g('bboxes'))                   # copies of the inputs: exact
        assert np.array_equal(r.bboxes_frame_inds.cpu().numpy(), g('frame_inds'))    # incl. the random shift
        assert np.allclose(r.iou.cpu().numpy(), g('iou'), atol=2e-5)                  # aligned IoU, HIP kernel
        assert np.array_equal(r.scores.cpu().numpy(), g('scores'))
        assert r.pos_gt_bboxes.shape == g('pos_gt_bboxes').shape                       # (0, 7) without a candidate
        assert np.array_equal(r.pos_gt_bboxes.cpu().numpy(), g('pos_gt_bboxes'))
        assert np.array_equal(r.pos_gt_labels.cpu().numpy(), g('pos_gt_labels'))
    assert np.array_equal(fi.cpu().numpy(), gold['assign_pts_frame_inds'])             # shifted in place, as upstream
    # the scene exercises what it claims to
    assert len(res[1].neg_inds) > 0 and len(res[3].pos_inds) == 0 and len(res[0].neg_inds) == 0


LOSS_KEYS = ['loss_rcnn_cls', 'num_pos_rois', 'num_neg_rois', 'loss_rcnn_bbox', 'num_occupied', 'num_free',
             'loss_rcnn_occ', 'recall_neg', 'recall_pos', 'precision_neg', 'precision_pos', 'acc', 'precision_posbox',
             'recall_posbox', 'precision_negbox', 'recall_negbox']


@pytest.mark.parametrize('split3', ['default', 'everywhere'])
def test_forward_train_losses_and_gradients_equal_reference(dev, gold, model, split3, monkeypatch):
    # 'everywhere': every f32 Linear of the model -- at the test scene's ~100 RoIs they would stay on the f32 library GEMM --
    # as one bf16 GEMM over three-way split operands (gemm.py, csrc/split3.hip: the default from 384 rows on): the SAME
    # tolerances against the imported reference's losses and gradients, which plain bf16 operands do not meet
    if split3 == 'everywhere':
        from objectcentricocccompletion_amd import gemm
        monkeypatch.setattr(gemm, 'SPLIT3_MIN_ROWS', 1)
        monkeypatch.setattr(gemm, 'SPLIT3_MIN_WORK', 0)
    samples, points, frames, trks, cands, occs, occ_scores = _scene(dev)
    torch.manual_seed(123)
    model.zero_grad(set_to_none=True)
    losses = model(return_loss=True, points=points, pts_frame_inds=[f.clone() for f in frames], img_metas=None,
                   tracklet=trks, gt_tracklet_candidates=cands, occ_labels=occs, occ_labels_scores=occ_scores)
    assert sorted(losses.keys()) == sorted(LOSS_KEYS)
    for k in LOSS_KEYS:
        got, exp = losses[k].detach().float().cpu().numpy().reshape(-1), gold['loss_' + k].reshape(-1)
        assert got.shape == exp.shape, k
        if k.startswith('num_'):
            assert np.array_equal(got, exp), k
        elif k == 'loss_rcnn_occ':                    # reduction='none': one BCE term per query point
            assert np.allclose(got, exp, rtol=1e-4, atol=2e-4), (k, np.abs(got - exp).max())
        else:
            assert np.allclose(got, exp, rtol=1e-4, atol=1e-5), (k, got, exp)
    total = losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()
    total.backward()
    params = dict(model.roi_head.bbox_head.named_parameters())
    names, norms = [str(n) for n in gold['grad_names']], gold['grad_norms']
    assert sorted(names) == sorted(params.keys())      # registration order differs, names do not
    worst = 0.0
    for n, e in zip(names, norms):
        g = params[n].grad
        got = 0.0 if g is None else float(g.double().norm())
        worst = max(worst, abs(got - e) / max(e, 1e-6 * float(norms.max())))
        assert abs(got - e) <= 2e-3 * e + 1e-6 * float(norms.max()), (n, got, e)
    for k in gold.files:
        if k.startswith('grad_') and k not in ('grad_names', 'grad_norms'):
            g, e = params[k[5:]].grad.cpu().numpy(), gold[k]
            assert np.abs(g - e).max() <= 2e-3 * np.abs(e).max() + 1e-7, (k, np.abs(g - e).max(), np.abs(e).max())
    model.zero_grad(set_to_none=True)
    print(f'worst relative gradient-norm error over {len(names)} parameters: {worst:.2e}')


@pytest.mark.parametrize('b', [0, 1, 2])
def test_simple_test_tracklets_and_occupancy_counts_equal_reference(dev, gold, model, b):
    samples, points, frames, trks, cands, occs, occ_scores = _scene(dev)
    t = trks[b]
    eye = torch.eye(4, device=dev)
    t.pose_list, t.shared_pose = [eye.clone() for _ in range(len(t))], eye.clone()
    before = t.boxes.clone()
    with torch.no_grad():
        r = model(return_loss=False, points=[points[b]], img_metas=[dict()], pts_frame_inds=[frames[b]], tracklet=[t],
                  gt_tracklet_candidates=[cands[b]], occ_labels=[occs[b]], occ_labels_scores=[occ_scores[b]])[0]
    assert sorted(r.keys()) == ['gt_boxes', 'inters', 'out_tracklets', 'unions']
    ot = r['out_tracklets'][0]
    assert torch.equal(t.boxes, before)                                    # the proposal is refined on a copy
    assert np.allclose(ot.boxes.cpu().numpy(), gold[f'test_boxes_{b}'], rtol=1e-4, atol=2e-4)
    assert np.allclose(ot.scores.cpu().numpy(), gold[f'test_scores_{b}'], rtol=1e-4, atol=1e-5)
    cat = lambda xs, empty: torch.cat(xs).numpy() if len(xs) else empty
    inters, unions = cat(r['inters'], np.zeros(0, np.int64)), cat(r['unions'], np.zeros(0, np.int64))
    assert inters.dtype == np.int64
    assert np.array_equal(inters, gold[f'test_inters_{b}'])               # integer counts: exact
    assert np.array_equal(unions, gold[f'test_unions_{b}'])
    assert np.allclose(cat(r['gt_boxes'], np.zeros((0, 7), np.float32)), gold[f'test_gt_boxes_{b}'], atol=1e-6)
    if b == 2:   # label confidence 0.3 < occ_label_thresh: nothing is counted
        assert len(r['inters']) == 0
    if b == 1:   # the candidate misses timestamps: those frames are not counted; frame 3 has no points and keeps its box
        assert len(inters) == len(samples[1]['candidates'][0][1])
        assert np.allclose(ot.boxes[3].cpu().numpy(), samples[1]['boxes'][3], atol=1e-6)


def test_graphed_transformer_equals_eager(dev):
    """heads.run_encoder: the temporal transformer replayed as a HIP-graph pair (forward / backward) gives what the eager
    launches give -- output, input gradients and parameter gradients (dropout off: same kernels, same order)."""
    import copy
    from objectcentricocccompletion_amd import heads
    from objectcentricocccompletion_amd.occ.layers import SimpleEncoderLayer, TransformerEncoder
    torch.manual_seed(0)
    L_, B, D = 32, 4, 256
    enc = TransformerEncoder(SimpleEncoderLayer(D, 4, dim_feedforward=512, dropout=0.0, mlp_dropout=0.0), 2).to(dev).train()
    ref = copy.deepcopy(enc)
    feats = torch.randn(L_, B, D, device=dev)
    pos = torch.randn(L_, B, D, device=dev)
    mask = torch.triu(torch.ones(L_, L_, dtype=torch.bool, device=dev), diagonal=1)
    dy = torch.randn(L_, B, D, device=dev)
    outs = []
    for model, graphed in ((enc, True), (ref, False)):
        heads.GRAPH_TRANSFORMER = graphed
        try:
            for it in range(2):    # (the second call replays what the first captured)
                model.zero_grad(set_to_none=True)
                f, p = feats.clone().requires_grad_(True), pos.clone().requires_grad_(True)
                y = heads.run_encoder(model, f, p, mask)
                y.backward(dy)
            outs.append((y.detach().clone(), f.grad.clone(), p.grad.clone(), [q.grad.clone() for q in model.parameters()]))
        finally:
            heads.GRAPH_TRANSFORMER = True
    assert enc in heads._graphed_encoders and ref not in heads._graphed_encoders
    a, b = outs
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp(min=1e-30))
    assert rel(a[0], b[0]) < 1e-5 and rel(a[1], b[1]) < 1e-5 and rel(a[2], b[2]) < 1e-5
    for x, y in zip(a[3], b[3]):
        assert rel(x, y) < 1e-4


def test_graphed_head_tail_equals_eager(dev):
    """heads._HeadTail (latent fusion + fused feature + cls / reg MLPs behind the transformer) as a graph pair against
    the same module called eagerly (dropout off)."""
    import copy
    from torch import nn
    from objectcentricocccompletion_amd import heads
    from objectcentricocccompletion_amd.sst.sst_ops import build_mlp
    torch.manual_seed(1)
    R, D = 128, 256
    ln = dict(type='LN', eps=1e-3)

    class Stub(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv_latent = build_mlp(2 * D, [D, D], ln, act='gelu')
            self.conv_fused = build_mlp(2 * D, [D], ln, act='gelu')
            self.conv_cls = build_mlp(D, [D, 1], ln, is_head=True, act='gelu')
            self.conv_reg = build_mlp(D, [D, 7], ln, is_head=True, act='gelu')
            self.fused_mode, self.rcnn_trans = 'concat_residual', True

    a = Stub().to(dev).train()
    b = copy.deepcopy(a)
    xs = [torch.randn(R, D, device=dev) for _ in range(3)]
    ds = [torch.randn(R, D, device=dev), torch.randn(R, 1, device=dev), torch.randn(R, 7, device=dev)]
    outs = []
    for model, graphed in ((a, True), (b, False)):
        for it in range(2):
            model.zero_grad(set_to_none=True)
            ins = [x.clone().requires_grad_(True) for x in xs]
            res = heads.graphed_call(model, heads._HeadTail, tuple(ins), slot='tail') if graphed else heads._HeadTail(model)(*ins)
            assert res is not None
            sum((r * d).sum() for r, d in zip(res, ds)).backward()
        outs.append(([r.detach().clone() for r in res], [i.grad.clone() for i in ins if i.grad is not None],
                     [q.grad.clone() for q in model.parameters()]))
        assert len(outs[-1][1]) == 2   # (rcnn_trans: the per-frame cluster features are not an input of the tail)
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp(min=1e-30))
    for k in range(3):
        for x, y in zip(outs[0][k], outs[1][k]):
            assert rel(x, y) < 1e-4, k


def test_training_step_host_fast_paths_equal_plain_paths(dev, monkeypatch):
    """One whole configs[2] training step (train mode, every dropout probability set to zero so that both runs are
    deterministic) with the round-3 host-side machinery on -- SIR layers as one library call per direction with their own
    weight-gradient kernel and queued parameter sums, the temporal transformer and the head's tail replayed as graph
    pairs -- against the same step with all of it off (per-block autograd nodes sequenced from Python, library GEMMs for
    the weight gradients, eager launches): the same losses, the same gradients."""
    import copy
    from objectcentricocccompletion_amd import heads, point_mlp, point_pool, roi_head, sir  # noqa: F401
    from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg
    from objectcentricocccompletion_amd.registry import DETECTORS
    from objectcentricocccompletion_amd.synthetic import synthetic_training_batch
    cfg = ococcnet_model_cfg()
    cfg['train_cfg']['random_shift_frame_inds'] = False
    bh = cfg['roi_head']['bbox_head']
    for k in ('attn_dropout', 'cls_dropout', 'reg_dropout', 'latent_dropout', 'fusion_dropout', 'dropout'):
        bh[k] = 0
    bh['occ_ae_head']['occ_decoder']['occ_dropout'] = 0
    torch.manual_seed(0)
    fast = DETECTORS.build(cfg).to(dev).train()
    plain = copy.deepcopy(fast)
    batch = synthetic_training_batch(2, 32, pts_per_frame=48, occ_queries=128, seed=3, device=dev)

    def run(model):
        outs = None
        for _ in range(2):   # (the second step replays what the first captured)
            model.zero_grad(set_to_none=True)
            losses = model(return_loss=True, **batch)
            total = losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()
            total.backward()
            outs = ({k: v.detach().float().mean().item() for k, v in losses.items()},
                    {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        return outs

    monkeypatch.setattr(heads, 'GRAPH_TRANSFORMER', True)
    a = run(fast)
    assert fast.roi_head.bbox_head in heads._graphed_encoders and fast.roi_head.bbox_head.trans_enc in heads._graphed_encoders
    for mod, name in ((sir, 'NATIVE_LAYER'), (sir, 'WHOLE_LAYER_NODE'), (point_mlp, 'WGRAD_KERNEL'), (heads, 'GRAPH_TRANSFORMER')):
        monkeypatch.setattr(mod, name, False)
    b = run(plain)
    assert plain.roi_head.bbox_head not in heads._graphed_encoders
    for k in b[0]:
        assert abs(a[0][k] - b[0][k]) <= 1e-5 * max(1.0, abs(b[0][k])), (k, a[0][k], b[0][k])
    assert set(a[1]) == set(b[1])
    worst = 0.0
    for k in b[1]:
        ref = float(b[1][k].abs().max())
        err = float((a[1][k] - b[1][k]).abs().max())
        worst = max(worst, err / max(ref, 1e-12))
        assert err <= 2e-4 * max(ref, 1e-9), (k, err, ref)
    print('largest gradient deviation (relative to the tensor\'s largest entry):', worst)


@pytest.mark.parametrize('shift', [False, True])
def test_batched_assignment_equals_per_tracklet_loop(dev, gold, model, monkeypatch, shift):
    """roi_head._assign_and_sample_batched (index lists of the whole batch in one upload, one gather per field, results as
    views) gives every field of every SamplingResult the per-tracklet loop gives (which the golden test above pins to the
    reference), including the frame-index shift and what it does to the points' frame indices."""
    from objectcentricocccompletion_amd import roi_head as rh
    samples, points, frames, trks, cands, occs, occ_scores = _scene(dev)
    batch_idx = torch.cat([torch.full((len(p),), i, dtype=torch.long, device=dev) for i, p in enumerate(points)])
    monkeypatch.setitem(model.roi_head.train_cfg, 'random_shift_frame_inds', shift)
    monkeypatch.setitem(model.roi_head.train_cfg, 'keep_frame_inds', False)
    outs = []
    for batched in (True, False):
        monkeypatch.setattr(rh, 'BATCHED_ASSIGN', batched)
        fi = torch.cat(frames).clone()
        torch.manual_seed(123)
        res = model.roi_head._assign_and_sample(trks, cands, occs, occ_scores, batch_idx, fi)
        outs.append((res, fi))
    (a, fa), (b, fb) = outs
    assert torch.equal(fa, fb)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        for name in ('pos_inds', 'neg_inds', 'pos_bboxes', 'neg_bboxes', 'pos_assigned_gt_inds', 'pos_gt_bboxes', 'pos_gt_labels',
                     'iou', 'scores', 'bboxes_frame_inds', 'bboxes'):
            u, v = getattr(x, name), getattr(y, name)
            assert u.dtype == v.dtype and u.shape == v.shape and torch.equal(u, v), name
        assert x.num_gts == y.num_gts and x.occ_labels is y.occ_labels and x.occ_scores is y.occ_scores
