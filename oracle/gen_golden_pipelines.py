"""Generate tests/golden/pipelines.npz: the REFERENCE's in-memory tracklet pipeline transforms
(mmdet3d/datasets/pipelines/tracklet_pipelines.py: TrackletNoise, PointDecoration, TrackletRandomFlip,
TrackletGlobalRotScaleTrans, TrackletRegularization, FrameDropout) run on a seeded tracklet with fixed RNG seeds,
imported through oracle/ref_shim.py in the build container only.  Data only, no reference source."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_shim as R  # noqa: E402
from oracle.gen_golden_tta import load_tracklet_classes, make_boxes  # noqa: E402


def load_pipelines():
    Boxes, Trk = load_tracklet_classes()
    pts = sys.modules['mmdet3d.core.points']
    base = R.load('mmdet3d.core.points.base_points')
    lidar = R.load('mmdet3d.core.points.lidar_points')
    pts.BasePoints, pts.LiDARPoints = base.BasePoints, lidar.LiDARPoints
    pts.get_points_type = lambda name: lidar.LiDARPoints
    # registry / base classes the pipeline module imports at the top
    reg = types.SimpleNamespace(register_module=lambda *a, **k: (lambda c: c))
    import mmdet.datasets.builder  # noqa (fabricated by the shim's auto-stub finder)
    import mmdet.datasets.pipelines  # noqa
    sys.modules['mmdet.datasets.builder'].PIPELINES = reg
    for n in ('LoadAnnotations', 'LoadImageFromFile'):
        setattr(sys.modules['mmdet.datasets.pipelines'], n, object)
    sys.modules['mmdet.datasets.pipelines'].to_tensor = torch.as_tensor
    core = sys.modules['mmdet3d.core']
    core.LiDARInstance3DBoxes = Boxes
    for name in ('mmdet3d.datasets.pipelines',):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(R.REF, *name.split('.'))]
            sys.modules[name] = m
    ds = sys.modules.get('mmdet3d.datasets') or types.ModuleType('mmdet3d.datasets')
    ds.__path__ = [os.path.join(R.REF, 'mmdet3d', 'datasets')]
    sys.modules['mmdet3d.datasets'] = ds
    tp = R.load('mmdet3d.datasets.pipelines.tracklet_pipelines')
    return Boxes, Trk, lidar.LiDARPoints, tp


def main():
    Boxes, Trk, LiDARPoints, tp = load_pipelines()
    g = torch.Generator().manual_seed(33)
    L = 7
    boxes = make_boxes(g, L)
    scores = (torch.rand(L, generator=g) * 0.8 + 0.1)
    npts = [int(v) for v in torch.randint(5, 40, (L,), generator=g)]
    points = [torch.cat([boxes[i:i + 1, :3] + torch.randn(n, 3, generator=g), torch.rand(n, 2, generator=g)], 1)
              for i, n in enumerate(npts)]
    out = dict(boxes=boxes.numpy(), scores=scores.numpy(), npts=np.array(npts),
               points=torch.cat(points, 0).numpy())

    def fresh():
        t = Trk('seg', 'id0', 1, False, box_list=[Boxes(boxes[i:i + 1].clone()) for i in range(L)],
                ts_list=list(range(100, 100 + L)), score_list=[float(s) for s in scores])
        t.pose_list = [torch.eye(4) for _ in range(L)]
        t.shared_pose = torch.eye(4)
        t.freeze()
        d = dict(tracklet=t, points=[p.clone() for p in points],
                 pts_frame_inds=[torch.ones(len(p), dtype=torch.int) * i for i, p in enumerate(points)])
        return d

    def cat_boxes(t):
        return torch.cat([b.tensor for b in t.box_list], 0).numpy().copy()

    # TrackletNoise (torch RNG)
    for consistent in (False, True):
        d = fresh()
        torch.manual_seed(7)
        tp.TrackletNoise(center_noise_cfg=dict(max_noise=[0.2, 0.2, 0.1], consistent=consistent),
                         size_noise_cfg=dict(max_noise=[0.2, 0.2, 0.1], consistent=consistent),
                         # (the reference's consistent yaw noise adds a [1] tensor to a scalar element: raises on this torch)
                         yaw_noise_cfg=dict(max_noise=0.2, consistent=False))(d)
        out[f'noise_{int(consistent)}'] = cat_boxes(d['tracklet'])
    # PointDecoration
    d = fresh()
    tp.PointDecoration(properties=['yaw', 'size', 'score', 'center_offset', 'length'], concat=False)(d)
    out['decorated'] = torch.cat(d['points'], 0).numpy()
    # flip + global rot / scale / trans on concatenated LiDARPoints (numpy RNG)
    d = fresh()
    allp = torch.cat(d['points'], 0)
    d['points'] = LiDARPoints(allp, points_dim=allp.shape[-1], attribute_dims=None)
    cand = Trk('seg', 'gt', 1, False, box_list=[Boxes(boxes[i:i + 1].clone() + 0.1) for i in range(L)],
               ts_list=list(range(100, 100 + L)), score_list=[1.0] * L)
    cand.shared_pose = torch.eye(4)
    cand.freeze()
    d['gt_tracklet_candidates'] = [cand]
    np.random.seed(11)
    tp.TrackletRandomFlip(flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5)(d)
    tp.TrackletGlobalRotScaleTrans(rot_range=[-0.78539816, 0.78539816], scale_ratio_range=[0.95, 1.05],
                                   translation_std=[0, 0, 0.2])(d)
    out['aug_points'] = d['points'].tensor.numpy().copy()
    out['aug_boxes'] = cat_boxes(d['tracklet'])
    out['aug_cand'] = cat_boxes(d['gt_tracklet_candidates'][0])
    out['aug_meta'] = np.array([float(d['pcd_horizontal_flip']), float(d['pcd_vertical_flip']), d['pcd_rot_angle'],
                                d['pcd_scale_factor'], *np.asarray(d['pcd_trans'], dtype=np.float64)])
    # TrackletRegularization: cut (numpy RNG) and pad
    d = fresh()
    np.random.seed(3)
    import warnings
    tp.TrackletRegularization(reg_len=4)(d)
    out['reg_cut_boxes'] = cat_boxes(d['tracklet'])
    out['reg_cut_ts'] = np.array(d['tracklet'].ts_list)
    out['reg_cut_npts'] = np.array([len(p) for p in d['points']])
    d = fresh()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        tp.TrackletRegularization(reg_len=10)(d)
    out['reg_pad_boxes'] = cat_boxes(d['tracklet'])
    out['reg_pad_npts'] = np.array([len(p) for p in d['points']])
    # FrameDropout
    d = fresh()
    np.random.seed(5)
    tp.FrameDropout(drop_ratio=0.45)(d)
    out['drop_ts'] = np.array(d['tracklet'].ts_list)
    out['drop_npts'] = np.array([len(p) for p in d['points']])
    # TrackletPoseTransform: per-frame ego poses (a drive along a curve), GT candidate with its own poses
    def pose(i):
        a = 0.05 * i
        m = torch.eye(4, dtype=torch.float64)
        m[:3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        m[:3, 3] = torch.tensor([2.0 * i, 0.3 * i * i, 0.01 * i])
        return m.float()
    for centering in (False, True):
        d = fresh()
        d['tracklet'].pose_list = [pose(i) for i in range(L)]
        d['tracklet'].shared_pose = None
        cand = Trk('seg', 'gt', 1, False, box_list=[Boxes(boxes[i:i + 1].clone() + 0.1) for i in range(L)],
                   ts_list=list(range(100, 100 + L)), score_list=[1.0] * L)
        cand.pose_list = [pose(i) for i in range(L)]
        cand.freeze()
        d['gt_tracklet_candidates'] = [cand]
        tp.TrackletPoseTransform(concat=False, centering=centering)(d)
        out[f'pose_{int(centering)}_points'] = torch.cat(d['points'], 0).numpy()
        out[f'pose_{int(centering)}_boxes'] = cat_boxes(d['tracklet'])
        out[f'pose_{int(centering)}_cand'] = cat_boxes(cand)
    out['poses'] = torch.stack([pose(i) for i in range(L)], 0).numpy()
    # ---- occupancy-label transforms (occ_pinelines.py): the module imports dataset base classes at the top
    for name, attrs in (('mmdet3d.datasets.pipelines', ('LoadPointsFromFile',)),
                        ('mmdet3d.datasets.pipelines.formating', ('DefaultFormatBundle3D',)),
                        ('mmdet3d.datasets.pipelines.transforms_3d', ('ObjectNameFilter', 'ObjectRangeFilter', 'RandomFlip3D'))):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for a in attrs:
            setattr(m, a, object)
    op = R.load('mmdet3d.datasets.pipelines.occ_pinelines')
    gg = torch.Generator().manual_seed(99)
    grids = [torch.randint(0, 3, (12, 9, 7), generator=gg), torch.randint(0, 3, (24, 10, 9), generator=gg),
             torch.zeros(5, 5, 5, dtype=torch.long), (torch.rand(6, 4, 3, generator=gg) < 0.3).long() * 2]
    for i, gr in enumerate(grids):
        out[f'occ_grid_{i}'] = gr.numpy()
    infos = [dict(occ_label_name=f'obj{i}') for i in range(len(grids))]

    def occ_results():
        return dict(occ_infos=infos, occ_label_list=[gr.clone() for gr in grids], occ_scores=torch.tensor([1.0, 0.7, 0.0, 0.5]))
    cfgs = dict(balance=dict(num_sample_points=64, pos_sample_weight=0.5, balance_sample=True, weighted_sample=True),
                weighted=dict(num_sample_points=64, pos_sample_weight=0.7, balance_sample=False, weighted_sample=True),
                plain=dict(num_sample_points=600, balance_sample=False, weighted_sample=False),
                unknown=dict(num_sample_points=64, use_unknown=True, balance_sample=False, weighted_sample=True),
                mirror=dict(num_sample_points=64, mirror_x=True, balance_sample=True),
                potential=dict(num_sample_points=64, use_potential=True, balance_sample=False, weighted_sample=False))
    for name, cfg in cfgs.items():
        d = occ_results()
        torch.manual_seed(13)
        t = op.RandomSampleOccPoints(voxel_size=0.2, **cfg)
        t(d)
        if name == 'potential':
            t(d)   # second call: the potentials of the first call steer the choice
        out[f'occ_{name}_labels'] = d['sample_occs'].numpy()
        out[f'occ_{name}_centers'] = d['sample_occ_centers'].numpy()
        out[f'occ_{name}_sizes'] = d['occ_sizes'].numpy()
        out[f'occ_{name}_scores'] = d['occ_scores'].numpy()
    d = occ_results()
    op.RandomSampleOccPoints(voxel_size=0.2, num_sample_points=-1)(d)
    for i in range(len(grids)):
        out[f'occ_all_labels_{i}'] = d['sample_occs'][i].numpy()
        out[f'occ_all_centers_{i}'] = d['sample_occ_centers'][i].numpy()
    d = occ_results()
    op.MirrorOccLabel()(d)
    for i in range(len(grids)):
        out[f'occ_mirrored_{i}'] = d['occ_label_list'][i].numpy()
    torch.manual_seed(2)
    dj = dict(sample_occ_centers=torch.from_numpy(out['occ_balance_centers']).clone())
    op.JitterOccCenter(voxel_size=0.2)(dj)
    out['occ_jittered'] = dj['sample_occ_centers'].numpy()
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'pipelines.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
