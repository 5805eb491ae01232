"""GT-occupancy annotation, device side: the range-image visibility ray test of
tools/occ/occ_annotate.py (point_cloud_to_range_image_idx :141-207, OccAnnotator.annotate_trk :488-556)
as one HIP kernel (ococc_occ_visibility_f64).  The file handling around it (Waymo frames, tracklet
aggregation, .npz writing) is host code outside this path."""
import ctypes
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
