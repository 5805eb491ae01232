"""the tracklet dataset and the full ococcnet.py train pipeline on files in the reference's
on-disk formats (written by tools/make_synthetic_dataset.py): proposals / GT-candidate pickles in LiDARTracklet dump
format, poses.pkl, per-tracklet point .npy, per-candidate occupancy .npz (waymo_tracklet_dataset.py:31-290, 491-584,
configs/ococc/ococcnet.py:183-262)."""
import importlib.util
import os
import pickle

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG_LEN = 32


@pytest.fixture(scope='module')
def data_root(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('synth_waymo'))
    spec = importlib.util.spec_from_file_location('make_synth', os.path.join(ROOT, 'tools', 'make_synthetic_dataset.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.main([root, '--tracklets', '5', '--frames', '40', '--seed', '1'])
    return root


def _pipeline():
    # configs/ococc/ococcnet.py:183-262 (train_pipeline), the steps up to the collecting bundle
    return [dict(type='LoadTrackletPoints', load_dim=6, use_dim=5, max_points=1024),
            dict(type='LoadTrackletAnnotations'),
            dict(type='LoadAnnotationsOcc', compute_score=False),
            dict(type='RandomSampleOccPoints', num_sample_points=512, pos_sample_weight=0.5, voxel_size=0.2,
                 use_unknown=False, use_potential=False, balance_sample=True, weighted_sample=True),
            dict(type='TrackletRegularization', reg_len=REG_LEN),
            dict(type='TrackletPoseTransform', concat=False),
            dict(type='TrackletNoise', center_noise_cfg=dict(max_noise=[0.2, 0.2, 0.1], consistent=False),
                 size_noise_cfg=dict(max_noise=[0.2, 0.2, 0.1], consistent=False),
                 yaw_noise_cfg=dict(max_noise=0.2, consistent=False)),
            dict(type='PointDecoration', properties=['yaw', 'size', 'score'], concat=True),
            dict(type='TrackletRandomFlip', flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
            dict(type='TrackletGlobalRotScaleTrans', rot_range=[-0.78539816, 0.78539816],
                 scale_ratio_range=[0.95, 1.05], translation_std=[0, 0, 0.2]),
            dict(type='PointsRangeFilter', point_cloud_range=[-204.7, -204.7, -3.99, 204.7, 204.7, 7.99]),
            dict(type='PointShuffle')]


def test_dump_format_round_trip(data_root):
    from objectcentricocccompletion_amd.tracklet import Tracklet
    items = pickle.load(open(os.path.join(data_root, 'tracklet_data', 'synth_training.pkl'), 'rb'))
    t = Tracklet.from_dump_format(items[0])
    assert len(t) == 40 and t.boxes.shape == (40, 7) and t.type == 1 and t.segment_name == 'segment-000'
    back = t.to_dump_format()
    assert back[:4] == items[0][:4] and back[5] == items[0][5]
    assert np.allclose(np.concatenate(back[4], 0), np.concatenate(items[0][4], 0))
    assert np.allclose(back[6], items[0][6], atol=1e-6)


def test_dataset_with_full_train_pipeline(data_root):
    from objectcentricocccompletion_amd import dataset  # noqa: F401 (registers)
    from objectcentricocccompletion_amd.registry import DATASETS
    ds = DATASETS.build(dict(type='WaymoTrackletDatasetWithOcc', data_root=data_root,
                             ann_file=os.path.join(data_root, 'tracklet_data', 'synth_training_gt_candidates.pkl'),
                             tracklet_proposals_file=os.path.join(data_root, 'tracklet_data', 'synth_training.pkl'),
                             occ_anno_root=os.path.join(data_root, 'occ_gt'), pose_file=os.path.join(data_root, 'poses.pkl'),
                             pipeline=_pipeline(), classes=['Car'], min_tracklet_points=100, min_tracklet_length=REG_LEN))
    assert len(ds) == 5
    np.random.seed(0)
    torch.manual_seed(0)
    s = ds[2]
    trk = s['tracklet']
    assert len(trk) == REG_LEN and trk.type == 0 and trk.type_name == 'Car'          # class id of the mmdet3d side
    assert s['points'].shape[1] == 10 and s['points'].shape[0] == s['pts_frame_inds'].shape[0]
    assert int(s['pts_frame_inds'].max()) == REG_LEN - 1 and int(s['pts_frame_inds'].min()) == 0
    assert s['sample_occs'].shape == (2, 512) and s['sample_occ_centers'].shape == (2, 512, 3)
    assert set(np.unique(s['sample_occs'].numpy())) <= {1, 2}                         # unknown cells are not sampled
    assert len(s['gt_tracklet_candidates']) == 2 and all(len(c) == 40 for c in s['gt_tracklet_candidates'])
    assert trk.shared_pose is not None and 'pcd_rot_angle' in s and 'pcd_horizontal_flip' in s
    # the augmentations moved points and boxes together: every point still lies inside (a slightly grown copy of)
    # its frame's box
    from objectcentricocccompletion_amd.bbox import rotation_3d_in_axis
    b = trk.boxes[s['pts_frame_inds'].long()]
    local = s['points'][:, :3] - b[:, :3]
    local[:, 2] -= b[:, 5] / 2
    local = rotation_3d_in_axis(local[:, None, :], -b[:, 6], axis=2)[:, 0]
    inside = (local.abs() <= b[:, 3:6] / 2 * 1.6 + 0.5).all(1)
    assert float(inside.float().mean()) > 0.9                 # (TrackletNoise moved the boxes, not the points)
    # short tracklets are filtered out
    ds2 = DATASETS.build(dict(type='WaymoTrackletDatasetWithOcc', data_root=data_root,
                              ann_file=os.path.join(data_root, 'tracklet_data', 'synth_training_gt_candidates.pkl'),
                              tracklet_proposals_file=os.path.join(data_root, 'tracklet_data', 'synth_training.pkl'),
                              occ_anno_root=os.path.join(data_root, 'occ_gt'), pose_file=os.path.join(data_root, 'poses.pkl'),
                              pipeline=_pipeline(), classes=['Car'], min_tracklet_length=41))
    assert len(ds2) == 0


def test_dump_format_equals_the_reference_tracklet_class(golden_dir):
    """tests/golden/formats.npz (oracle/gen_golden_formats.py): the tuple the REFERENCE's LiDARTracklet.to_dump_format
    wrote and what its from_dump_format read back -- Tracklet.from_dump_format reads that tuple into the same boxes /
    timestamps / scores, and Tracklet.to_dump_format writes it back field for field."""
    from objectcentricocccompletion_amd.tracklet import Tracklet
    G = np.load(os.path.join(golden_dir, 'formats.npz'))
    assert int(G['tuple_len']) == 8 and tuple(G['box_shape']) == (1, 7)
    item = (str(G['segment_name']), str(G['id']), int(G['type']), bool(G['in_world']), [G['boxes'][i:i + 1] for i in range(len(G['boxes']))],
            [int(t) for t in G['ts']], [float(s) for s in G['scores']], [int(n) for n in G['num_pts']])
    t = Tracklet.from_dump_format(item)
    assert len(t) == int(G['back_len']) == int(G['back_size']) and t.ts_list == [int(v) for v in G['back_ts']]
    assert np.array_equal(t.boxes.numpy(), G['back_boxes']) and t.boxes.dtype == torch.float32
    assert (t.segment_name, t.id, t.type, t.in_world) == (item[0], item[1], item[2], item[3])
    back = t.to_dump_format()
    assert len(back) == 8 and back[:4] == item[:4] and back[5] == item[5] and back[7] == item[7]
    assert all(b.shape == (1, 7) and b.dtype == np.float32 for b in back[4])
    assert np.array_equal(np.concatenate(back[4], 0), G['boxes'])
    assert np.allclose(back[6], item[6], rtol=0, atol=1e-7)      # scores: python floats there, float32 tensor here
    # and the pickle of the reference's tuple loads as is (protocol 2 and up)
    assert pickle.loads(pickle.dumps(item, protocol=2))[5] == item[5]
