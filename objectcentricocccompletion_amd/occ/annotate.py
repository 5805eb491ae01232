"""GT-occupancy annotation, device side: the range-image visibility ray test of
tools/occ/occ_annotate.py (point_cloud_to_range_image_idx :141-207, OccAnnotator.annotate_trk :488-556)
as one HIP kernel (ococc_occ_visibility_f64), and the file handling around it (OccAnnotator below: tracklets from the
GT metrics file, per-frame points and raw range images, aggregation in the box frame, voxelisation, the .npz files
LoadAnnotationsOcc reads)."""
import math

import numpy as np
import torch

from .. import _lib as L


def _affines_to_sensor(extrinsics):
    """[S,4,4] LiDAR extrinsics (sensor -> vehicle) -> [S,12] vehicle -> sensor (rotation row-major, translation)
    and the azimuth correction atan2(e[1,0], e[0,0]) (occ_annotate.py:158-178)."""
    e = np.asarray(extrinsics, dtype=np.float64)
    inv = np.linalg.inv(e)
    aff = np.concatenate([inv[:, :3, :3].reshape(-1, 9), inv[:, :3, 3]], 1)
    return aff, np.arctan2(e[:, 1, 0], e[:, 0, 0])


def frame_affines_from_boxes(boxes):
    """[F,7] tracklet boxes (float32, as LiDARInstance3DBoxes holds them) -> [F,12] object -> ego affines the way
    annotate_trk builds them (:497-509): sin / cos of the float32 yaw, widened to float64; p_ego = p @ rot_T + origin."""
    b = torch.as_tensor(boxes, dtype=torch.float32).cpu()
    s, c = torch.sin(b[:, 6]).double().numpy(), torch.cos(b[:, 6]).double().numpy()
    o = b[:, :3].double().numpy()
    z, one = np.zeros_like(s), np.ones_like(s)
    # rot_T = [[c,-s,0],[s,c,0],[0,0,1]] applied to row vectors  ==  R = rot_T^T applied to columns
    rot = np.stack([c, s, z, -s, c, z, z, z, one], 1)
    return np.concatenate([rot, o], 1)


def _run(centers, to_ego, to_sensor, az_corr, inclinations, range_images, size, want_vis, want_dbg):
    dev = centers.device
    L.require_device(centers)
    c = centers.to(torch.float64).contiguous()
    n = c.size(0)
    frames = to_ego.shape[0]
    sf = to_sensor.shape[0]
    assert sf % frames == 0
    height, width = int(size[0]), int(size[1])
    t_ego = torch.from_numpy(np.ascontiguousarray(to_ego, dtype=np.float64)).to(dev)
    t_sen = torch.from_numpy(np.ascontiguousarray(to_sensor, dtype=np.float64)).to(dev)
    azc = torch.from_numpy(np.ascontiguousarray(az_corr, dtype=np.float64)).to(dev)
    inc = torch.as_tensor(inclinations, dtype=torch.float64).reshape(sf, height).contiguous().to(dev)
    vis = torch.empty((n,), dtype=torch.int32, device=dev) if want_vis else None
    idx = torch.empty((sf, n, 2), dtype=torch.int32, device=dev) if want_dbg else None
    rng = torch.empty((sf, n), dtype=torch.float64, device=dev) if want_dbg else None
    ptrs, keep, code = None, [], L.F32
    if want_vis:
        assert len(range_images) == sf
        dt = range_images[0].dtype
        assert dt in (torch.float32, torch.float64) and all(r.dtype == dt for r in range_images)
        code = 2 if dt == torch.float64 else L.F32
        keep = [r.to(dev).contiguous() for r in range_images]
        assert all(tuple(r.shape) == (height, width) for r in keep)
        table = torch.tensor([r.data_ptr() for r in keep], dtype=torch.int64, device=dev)
        keep.append(table)
        ptrs = table.data_ptr()
    L.check(L.lib.ococc_occ_visibility_f64(L.ptr(c), n, L.ptr(t_ego), frames, L.ptr(t_sen), L.ptr(azc), L.ptr(inc),
                                           sf // frames, height, width, ptrs, code, L.ptr(vis), L.ptr(idx), L.ptr(rng),
                                           L.stream()), 'occ_visibility')
    if keep:
        torch.cuda.current_stream().synchronize()  # the pointer table and the images stay alive until the launch ran
    return vis, idx, rng


def point_cloud_to_range_image_idx(points, extrinsics, inclinations, range_image_size):
    """Same arguments and results as tools/occ/occ_annotate.py:141-207: points [B,N,3] (vehicle frame, one point
    set per frame), extrinsics [B,4,4], inclinations [B,H] -> (ri_indices [B,N,2] int32 (row, col), ri_range [B,N]
    float64).  One launch per frame (each frame has its own points here); the annotation itself uses
    visibility_ray_test, which shares one set of cell centres over all frames and sensors."""
    B = points.shape[0]
    to_sensor, azc = _affines_to_sensor(extrinsics.detach().cpu().numpy())
    ident = np.concatenate([np.eye(3).reshape(1, 9), np.zeros((1, 3))], 1)
    inds, rngs = [], []
    for b in range(B):
        _, idx, rng = _run(points[b], ident, to_sensor[b:b + 1], azc[b:b + 1], inclinations[b:b + 1], None,
                           range_image_size, False, True)
        inds.append(idx[0])
        rngs.append(rng[0])
    return torch.stack(inds, 0), torch.stack(rngs, 0)


def visibility_ray_test(unknown_centers, track_boxes, extrinsics, inclinations, range_images):
    """Label the unoccupied cell centres of one object grid (annotate_trk :488-556).

    unknown_centers [N,3] object-frame centres; track_boxes [F,7] the tracklet's boxes (object -> ego per frame);
    extrinsics [S,F,4,4], inclinations [S,F,H] (already in range-image row order), range_images: S lists of F
    [H,W] tensors.  Returns visibility [N] int32: 2 = some ray crossed the cell (empty), 0 = never seen."""
    ext = np.asarray(extrinsics, dtype=np.float64)
    S, F = ext.shape[:2]
    to_sensor, azc = _affines_to_sensor(ext.reshape(S * F, 4, 4))
    imgs = [range_images[s][f] for s in range(S) for f in range(F)]
    size = tuple(imgs[0].shape)
    vis, _, _ = _run(unknown_centers, frame_affines_from_boxes(track_boxes), to_sensor, azc,
                     torch.as_tensor(np.asarray(inclinations, dtype=np.float64)).reshape(S * F, -1), imgs, size, True, False)
    return vis


# ------------------------------------------------------------------------------------------------------------------
# File handling around the ray test: tools/occ/occ_annotate.py:103-139 (get_local_point_list), :228-688 (OccAnnotator)
# and the readers of tools/ctrl/utils.py it uses, on the same directory layout:
#   <data_root>/kitti_format/idx2timestamp.pkl            {frame index (str): timestamp}
#   <data_root>/kitti_format/<split>/velodyne/<idx>.bin   float32 [N, 6] points of the frame (get_pc_from_time_stamp)
#   <data_root>/waymo_raw/<split>/<idx>.pkl               per frame: '<LIDAR>_BEAM_INCLINATION' [H], '<LIDAR>_LIDAR_EXTRINSIC'
#                                                         [4,4], '<LIDAR>_RANGE_IMAGE_MERGE_VIRTUAL' [H,W] for the five LiDARs
#   <bin_path>                                            metrics.Objects file of the GT boxes (waymo_io.read_bin)
# and writes <out_dir>/<split>/<segment>/<track id>.npz with key 'occ' [X,Y,Z] int32: 0 unknown, 1 occupied, 2 empty
# (what LoadAnnotationsOcc reads).  The Open3D voxelisation branch (--cpu-voxelization) needs open3d, which neither the
# reference's requirements nor this image carry: NotImplementedError.  One process per GPU: a multi-GPU run shards the
# segments by rank (``rank`` / ``world``) instead of the reference's worker pool over cuda devices.
def points_in_box(points, box):
    """check_pt_in_box3d of mmdet3d/ops/roiaware_pool3d/src/points_in_boxes_cuda.cu:24-49 for ONE box [7] (x, y, z bottom,
    w, l, h, rz), float32 arithmetic: bool [N]."""
    p = points[:, :3].float()
    cx, cy, cz, w, l, h, rz = [box[i].float() for i in range(7)]
    cz = cz + h / 2.0
    rot = rz + math.pi / 2
    cosa, sina = torch.cos(rot), torch.sin(rot)
    dx, dy = p[:, 0] - cx, p[:, 1] - cy
    lx, ly = dx * cosa + dy * (-sina), dx * sina + dy * cosa
    return ~((p[:, 2] - cz).abs() > h / 2.0) & (lx > -l / 2.0) & (lx < l / 2.0) & (ly > -w / 2.0) & (ly < w / 2.0)


def get_local_point_list(trk, load_points, box_mode='avg', device='cuda'):
    """occ_annotate.py:103-139: the points inside every frame's box, moved into the box frame (origin at the bottom centre,
    x along the heading), and the aggregate box size.  load_points(ts) -> float32 [N, >=3] ndarray."""
    local_pc_list, box_sizes = [], []
    boxes = trk.boxes.to(device).float()
    for i in range(len(trk)):
        pc = torch.from_numpy(np.ascontiguousarray(load_points(trk.ts_list[i])[:, :3])).to(device)
        inside = pc[points_in_box(pc, boxes[i])]
        if len(inside) == 0:
            continue
        rz = boxes[i, 6]
        c, s = torch.cos(-rz), torch.sin(-rz)
        rot_mat_T = torch.stack([torch.stack([c, -s, c.new_zeros(())]), torch.stack([s, c, c.new_zeros(())]),
                                 c.new_tensor([0., 0., 1.])])
        local_pc_list.append((inside - boxes[i, :3]) @ rot_mat_T)
        box_sizes.append(boxes[i:i + 1, 3:6])
    assert len(local_pc_list) > 0, 'no points in the tracklet'
    sizes = torch.cat(box_sizes, 0)
    return local_pc_list, (sizes.mean(0) if box_mode == 'avg' else sizes.max(0).values)


class OccAnnotator(object):
    """occ_annotate.py:228-688 with the reference's constructor arguments (``workers`` / ``ngpus`` are accepted and
    replaced by ``rank`` / ``world``: one process per GPU)."""
    type_mapping = {'vehicle': 1, 'pedestrian': 2, 'cyclist': 3}
    LiDAR_NAME_LIST = ['TOP', 'FRONT', 'SIDE_LEFT', 'SIDE_RIGHT', 'REAR']

    def __init__(self, data_root, out_dir, split, voxel_size, bin_path, object_type='vehicle', workers=1, debug=False,
                 cpu_voxelization=False, overwrite=False, save_mean_var=False, ngpus=1, rank=0, world=1, device='cuda'):
        import os.path as osp
        import pickle
        if cpu_voxelization:
            raise NotImplementedError('the Open3D voxelisation branch needs open3d (absent here as in the reference\'s requirements)')
        self.data_root, self.out_dir, self.split, self.voxel_size = data_root, out_dir, split, voxel_size
        self.kitti_format_root = osp.join(data_root, 'kitti_format')
        self.raw_format_root = osp.join(data_root, 'waymo_raw', split)
        self.debug, self.overwrite, self.save_mean_var = debug, overwrite, save_mean_var
        self.rank, self.world, self.device = rank, world, device
        self.types = {self.type_mapping[object_type]}
        name = osp.basename(bin_path).split('.')[0]
        tracklets = self.generate_or_load_tracklet(bin_path, f'{name}_tracklets.pkl')
        self.trk_dicts = {}
        for t in tracklets:
            self.trk_dicts.setdefault(t.segment_name, []).append(t)
        self.segment_names = sorted(self.trk_dicts)
        with open(osp.join(self.kitti_format_root, 'idx2timestamp.pkl'), 'rb') as fr:
            self.idx2ts = pickle.load(fr)
        self.ts2idx = {ts: idx for idx, ts in self.idx2ts.items()}

    def generate_or_load_tracklet(self, bin_path, file_name):
        """:268-281 -- the tracklets of the metrics file, cached beside the outputs (as dump-format tuples: the reference
        pickles its LiDARTracklet objects, which only its own class can load)"""
        import os
        import pickle
        from .. import waymo_io
        from ..tracklet import Tracklet
        cache = os.path.join(self.out_dir, file_name)
        if os.path.isfile(cache):
            with open(cache, 'rb') as f:
                return [Tracklet.from_dump_format(t) for t in pickle.load(f)]
        tracklets = waymo_io.generate_tracklets(waymo_io.read_bin(bin_path), self.types)
        os.makedirs(self.out_dir, exist_ok=True)
        with open(cache, 'wb') as f:
            pickle.dump([t.to_dump_format() for t in tracklets], f)
        return tracklets

    def load_points(self, ts):
        """tools/ctrl/utils.py:60-66"""
        import os.path as osp
        path = osp.join(self.kitti_format_root, f'{self.split}/velodyne', str(self.ts2idx[ts]) + '.bin')
        return np.fromfile(path, dtype=np.float32).reshape(-1, 6)

    def annotate_one_seg(self, segname_idx):
        cache = {}

        def load(ts):
            if ts not in cache:
                cache[ts] = self.load_points(ts)
            return cache[ts]
        done = 0
        for trk in self.trk_dicts[self.segment_names[segname_idx]]:
            done += self.annotate_trk(trk, load) is not None
        return done

    def annotate_trk(self, trk, load_points=None):
        """:312-655 (the GPU voxelisation branch).  Returns the path written, or None when the tracklet is skipped (an
        existing readable file, fewer than 10 frames, no point in any box, a missing raw frame)."""
        import os
        import pickle
        dev = self.device
        out_path = os.path.join(self.out_dir, self.split, trk.segment_name)
        os.makedirs(out_path, exist_ok=True)
        out_name = os.path.join(out_path, f'{trk.id}.npz')
        if os.path.isfile(out_name) and not self.overwrite:
            try:
                np.load(out_name)
                return None
            except Exception:
                print(f'error loading {out_name}, overwrite')
        if len(trk) < 10:
            return None
        try:
            local_pc_list, bbox_size = get_local_point_list(trk, load_points or self.load_points, 'max', dev)
        except AssertionError as e:
            print(e)
            return None
        local_pc_agg = torch.cat(local_pc_list, 0)
        voxel_dims = torch.ceil(bbox_size / self.voxel_size).to(torch.int32)
        # corners of the local box (origin (0.5, 0.5, 0), yaw 0): x, y centred, z from the bottom face
        min_bound = torch.stack([-bbox_size[0] / 2, -bbox_size[1] / 2, bbox_size.new_zeros(())])
        quantized = torch.floor((local_pc_agg - min_bound) / self.voxel_size).to(torch.long)
        keep = (quantized < voxel_dims[None]).all(1) & (quantized >= 0).all(1)   # (points right on the boundary: dropped)
        local_pc_agg, quantized = local_pc_agg[keep], quantized[keep]
        X, Y, Z = (int(v) for v in voxel_dims)
        occ = torch.zeros((X, Y, Z), dtype=torch.bool, device=dev)
        occ[quantized[:, 0], quantized[:, 1], quantized[:, 2]] = True
        gx, gy, gz = torch.meshgrid(torch.arange(X, device=dev), torch.arange(Y, device=dev), torch.arange(Z, device=dev),
                                    indexing='ij')
        voxel_coors = torch.stack([gx, gy, gz], -1).view(-1, 3)
        occ = occ.view(-1)
        un_occ_coors = voxel_coors[~occ]
        unknown_centers = un_occ_coors.to(torch.float64) * self.voxel_size + min_bound + self.voxel_size / 2
        visible_occ = torch.zeros_like(occ, dtype=torch.int32)
        if un_occ_coors.size(0) > 0:
            frames = []
            for ts in trk.ts_list:
                path = os.path.join(self.raw_format_root, f'{self.ts2idx[ts]}.pkl')
                if not os.path.isfile(path):
                    print(f'{path} not found, skip this segment {trk.segment_name}')
                    return None
                try:
                    with open(path, 'rb') as f:
                        frames.append(pickle.load(f))
                except Exception:
                    print(f'error loading {path}, skip this segment {trk.segment_name}')
                    return None
            visibility = None
            for lidar in self.LiDAR_NAME_LIST:   # every LiDAR has its own image size: one ray test per sensor, then the max
                ext = np.stack([fr[f'{lidar}_LIDAR_EXTRINSIC'] for fr in frames], 0)[None]
                inc = np.flip(np.stack([fr[f'{lidar}_BEAM_INCLINATION'] for fr in frames], 0), axis=1).copy()[None]
                imgs = [[torch.as_tensor(np.asarray(fr[f'{lidar}_RANGE_IMAGE_MERGE_VIRTUAL'])) for fr in frames]]
                vis = visibility_ray_test(unknown_centers, trk.boxes, ext, inc, imgs)
                visibility = vis if visibility is None else torch.maximum(visibility, vis)
            visible_occ[~occ] = visibility
        visible_occ[occ] = 1
        grid = visible_occ.view(X, Y, Z)
        if self.save_mean_var:
            # per occupied cell the mean and the variance of its points (scatter_v2 'mean' twice, :633-651)
            flat = (quantized[:, 0] * Y + quantized[:, 1]) * Z + quantized[:, 2]
            cnt = torch.zeros(X * Y * Z, device=dev, dtype=local_pc_agg.dtype).index_add_(0, flat, torch.ones_like(flat, dtype=local_pc_agg.dtype))
            mean = torch.zeros((X * Y * Z, 3), device=dev, dtype=local_pc_agg.dtype).index_add_(0, flat, local_pc_agg)
            mean = mean / cnt.clamp(min=1)[:, None]
            var = torch.zeros_like(mean).index_add_(0, flat, (local_pc_agg - mean[flat]) ** 2) / cnt.clamp(min=1)[:, None]
            np.savez(out_name, occ=grid.cpu().numpy(), mean_var=torch.cat([mean, var], 1).view(X, Y, Z, 6).cpu().numpy())
        else:
            np.savez(out_name, occ=grid.cpu().numpy())
        return out_name

    def annotate_segment(self, chunksize=-1):
        """:657-688; segments rank, rank + world, ... of the sorted list"""
        n = 0
        for i in range(self.rank, len(self.segment_names), self.world):
            n += self.annotate_one_seg(i)
        return n
