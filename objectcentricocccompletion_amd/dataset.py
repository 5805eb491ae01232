"""Tracklet dataset with occupancy labels -- WaymoTrackletDatasetWithOcc and its base WaymoTrackletDataset
(mmdet3d/datasets/waymo_tracklet_dataset.py:31-290, 491-584) on the same files:

* tracklet_proposals_file (``*_training.pkl``): list of tracklet tuples in dump format (tracklet.Tracklet
  .from_dump_format) or of (tuple, points path, (beg, end)) triples;
* ann_file (``*_gt_candidates.pkl``): per proposal, the list of GT-candidate tracklet tuples;
* pose_file (``poses.pkl``): {timestamp: 4x4 ego -> world};
* <proposals file minus .pkl>_database/<segment>--<id>.npy: the tracklet's points (LoadTrackletPoints);
* occ_anno_root/<segment>/<id>.npz, key ``occ``: the GT occupancy grid of a candidate (LoadAnnotationsOcc).

``evaluate`` (:315-428) writes the Waymo-format result file (waymo_io.convert_tracklet_to_waymo: the metrics.Objects
protobuf encoded without the waymo_open_dataset package) and parses the metrics tool's output when the caller names the
tool's executable; the occupancy-IoU metric is roi_head.occupancy_iou_metrics."""
import os.path as osp
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

from .pipelines import Compose
from .registry import DATASETS
from .tracklet import Tracklet


def _load_pickle(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


@DATASETS.register_module()
class WaymoTrackletDataset(Dataset):
    CLASSES = ('Car', 'Pedestrian', 'Cyclist')

    def __init__(self, data_root, ann_file, tracklet_proposals_file, pose_file, pipeline=None, classes=None,
                 box_type_3d='LiDAR', test_mode=False, load_interval=1, min_tracklet_points=1):
        super().__init__()
        self.data_root, self.ann_file, self.test_mode = data_root, ann_file, test_mode
        self.CLASSES = tuple(classes) if classes is not None else self.CLASSES
        self.cat2id = {name: i for i, name in enumerate(self.CLASSES)}
        if ann_file is not None:
            self.ann_candidates = _load_pickle(ann_file)
        self.tracklet_proposals_file = tracklet_proposals_file
        if tracklet_proposals_file is not None:
            infos = _load_pickle(tracklet_proposals_file)
            # keep proposals with enough points in their boxes (last tuple field) of Waymo type 1 (vehicle), :95-99
            if len(infos[0]) <= 3:
                mask = [sum(e[0][-1]) >= min_tracklet_points and e[0][2] == 1 for e in infos]
            else:
                mask = [sum(e[-1]) >= min_tracklet_points and e[2] == 1 for e in infos]
            self.data_infos = [e for e, m in zip(infos, mask) if m][::load_interval]
            if hasattr(self, 'ann_candidates'):
                self.ann_candidates = [e for e, m in zip(self.ann_candidates, mask) if m][::load_interval]
        self.poses = {k: torch.from_numpy(np.asarray(p)).float() for k, p in _load_pickle(pose_file).items()}
        self.pipeline = Compose(pipeline) if pipeline is not None else None
        self.pipeline_types = [p['type'] for p in pipeline] if pipeline is not None else []
        self._skip_type_keys = None
        if not self.test_mode:
            self._set_group_flag()

    def __len__(self):
        return len(self.data_infos)

    def _set_group_flag(self):
        self.flag = np.zeros(len(self), dtype=np.uint8)

    def update_skip_type_keys(self, skip_type_keys):
        self._skip_type_keys = skip_type_keys

    def _tracklet(self, item):
        trk = Tracklet.from_dump_format(item)
        trk.set_poses(self.poses)
        trk.set_type_name()
        trk.set_type(self.cat2id[trk.type_name], 'mmdet3d')
        return trk

    def get_data_info(self, index):
        info = self.data_infos[index]
        specified_path = sub_interval = None
        if len(info) == 3:
            info, specified_path, sub_interval = info
        trk = self._tracklet(info)
        f = self.tracklet_proposals_file
        for tag in ('_static', '_dynamic'):
            if tag[1:] in f:
                assert tag + '.pkl' in f
                f = f.replace(tag, '')
        pts_filename = specified_path or osp.join(f.replace('.pkl', '_database'), trk.segment_name + '--' + trk.id + '.npy')
        out = dict(pts_filename=pts_filename, sample_idx=trk.id, file_name=pts_filename, tracklet=trk,
                   point_cloud_interval=sub_interval)
        if not self.test_mode:
            out['ann_info'] = self.get_ann_info(index)
        return out

    def get_ann_info(self, index):
        return [self._tracklet(t) for t in self.ann_candidates[index]]

    def prepare_train_data(self, index):
        example = self.get_data_info(index)
        if example is None:
            return None
        for transform, ttype in zip(self.pipeline.transforms, self.pipeline_types):
            if self._skip_type_keys is not None and ttype in self._skip_type_keys:
                continue
            example = transform(example)
        return example

    def prepare_test_data(self, index):
        return self.pipeline(self.get_data_info(index))

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_data(idx)
        while True:
            data = self.prepare_train_data(idx)
            if data is None:
                idx = np.random.choice(np.where(self.flag == self.flag[idx])[0])
                continue
            return data

    def evaluate(self, results, metric='waymo', logger=None, pklfile_prefix=None, submission_prefix=None, show=False,
                 out_dir=None, pipeline=None, metrics_main=None):
        """:315-428 -- results: the refined tracklets (``id`` str, ``segment_name``, ``type`` = class index); writes
        ``pklfile_prefix``.bin and evaluates it against <waymo_format>/gt.bin (train_gt.bin for 'result_train' prefixes)
        with the Waymo tool ``metrics_main`` (waymo_io.evaluate: without the tool, the .bin is written and the call
        raises)."""
        from . import waymo_io
        waymo_root = osp.join(self.data_root.split('kitti_format')[0], 'waymo_format')
        gt = osp.join(waymo_root, 'train_gt.bin' if 'result_train' in pklfile_prefix else 'gt.bin')
        return waymo_io.evaluate(results, pklfile_prefix, gt, self.CLASSES, metrics_main)


@DATASETS.register_module()
class WaymoTrackletDatasetWithOcc(WaymoTrackletDataset):
    """Adds the occupancy-label file of every GT candidate (:491-584)."""

    def __init__(self, data_root, ann_file, tracklet_proposals_file, occ_anno_root, pose_file, pipeline=None,
                 classes=None, box_type_3d='LiDAR', test_mode=False, load_interval=1, min_tracklet_length=50,
                 min_tracklet_points=1):
        super().__init__(data_root, ann_file, tracklet_proposals_file, pose_file, pipeline, classes, box_type_3d,
                         False, load_interval, min_tracklet_points)
        self.min_tracklet_length = min_tracklet_length
        if min_tracklet_length > 0:
            self.filter_tracklets_by_length()
        self.gt_anno_occ = True
        self.occ_anno_root = occ_anno_root
        # (the reference passes test_mode=False to its base class whatever the argument says, :519, so the candidates
        # and their occupancy files are always attached; kept)
        self._set_group_flag()

    def filter_tracklets_by_length(self):
        mask = [len((e[0] if len(e) == 3 else e)[-1]) >= self.min_tracklet_length for e in self.data_infos]
        self.data_infos = [e for e, m in zip(self.data_infos, mask) if m]
        if hasattr(self, 'ann_candidates'):
            self.ann_candidates = [e for e, m in zip(self.ann_candidates, mask) if m]

    def get_data_info(self, index):
        out = super().get_data_info(index)
        out['occ_infos'] = [self.parse_occ_anno(t) for t in out['ann_info']]
        return out

    def parse_occ_anno(self, trk):
        return dict(occ_label_name=osp.join(self.occ_anno_root, trk.segment_name, f'{trk.id}.npz'), label_iou=1.0,
                    label_trk_length=len(trk))
