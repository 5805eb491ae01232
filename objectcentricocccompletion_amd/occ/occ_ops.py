"""Host mirror of mmdet3d/ops/occ/occ_ops.py: voxel-centre helpers of the implicit
occupancy grid (generate_dense_voxel_centers :5-50, quantize_points :53-93,
jitter_voxel_center :96-100).  Elementwise index math on small tensors."""
import torch

from .._lib import const_tensor


def generate_dense_voxel_centers(bbox_sizes, voxel_size, scale_wlh=[1.0, 1.0, 1.0],
                                 offset_wlh=[0.0, 0.0, 0.0], as_volume=False):
    """Centres of the ceil(size / voxel) cells of each box, in the box frame (origin at the
    box centre), x fastest last: list of [X*Y*Z, 3] (or [X,Y,Z,3]) tensors."""
    out = []
    for size in bbox_sizes:
        size = size * size.new_tensor(scale_wlh) + size.new_tensor(offset_wlh)
        n = torch.ceil(size / voxel_size)
        xs, ys, zs = [int(v) for v in n.tolist()]
        dev = bbox_sizes.device
        gx, gy, gz = torch.meshgrid(torch.arange(xs, device=dev), torch.arange(ys, device=dev),
                                    torch.arange(zs, device=dev), indexing='ij')
        coors = torch.stack([gx, gy, gz], dim=-1).view(-1, 3)
        centers = coors.to(torch.float) * voxel_size + (-size / 2) + voxel_size / 2
        if as_volume:
            centers = centers.view(xs, ys, zs, 3)
        out.append(centers)
    return out


def quantize_points(points, rois, rois_points_idx, voxel_size, scale_wlh=[1.0, 1.0, 1.0],
                    offset_wlh=[0.0, 0.0, 0.0], to_center=False):
    """Voxel index floor((p + size/2) / voxel) of box-frame points, the volume centred on the
    (enlarged) RoI of each point; with to_center the centre of that voxel (occ_ops.py:53-93)."""
    sizes = rois[:, 4:7]
    sizes = sizes * const_tensor(scale_wlh, sizes.device, sizes.dtype).view(1, 3) \
        + const_tensor(offset_wlh, sizes.device, sizes.dtype).view(1, 3)
    min_bound = (-sizes / 2)[rois_points_idx.long()]
    voxel_coors = torch.floor((points - min_bound) / voxel_size).to(torch.long)
    if to_center:
        return voxel_coors.to(torch.float) * voxel_size + min_bound + voxel_size / 2
    return voxel_coors


def jitter_voxel_center(voxel_size, voxel_centers):
    return voxel_centers + torch.rand_like(voxel_centers) * voxel_size - voxel_size / 2


def dense_voxel_centers_batched(bbox_sizes, voxel_size, scale_wlh=[1.0, 1.0, 1.0], offset_wlh=[0.0, 0.0, 0.0]):
    """All boxes of generate_dense_voxel_centers in one flat tensor: (centers [sum K, 3], box index [sum K],
    K per box [R]).  Same cell order (x slowest, z fastest) and the same float expression per element, so
    ``centers[box == j]`` equals ``generate_dense_voxel_centers(...)[j]`` bit for bit; one host read-back
    (the total) instead of a Python loop over boxes."""
    dev = bbox_sizes.device
    if bbox_sizes.size(0) == 0:
        z = torch.zeros((0,), dtype=torch.long, device=dev)
        return bbox_sizes.new_zeros((0, 3)), z, z
    size = bbox_sizes * bbox_sizes.new_tensor(scale_wlh) + bbox_sizes.new_tensor(offset_wlh)     # [R,3]
    dims = torch.ceil(size / voxel_size).to(torch.long)                                             # [R,3]
    k = dims[:, 0] * dims[:, 1] * dims[:, 2]
    total = int(k.sum())
    start = torch.cumsum(k, 0) - k
    box = torch.repeat_interleave(torch.arange(size.size(0), device=dev), k, output_size=total)
    local = torch.arange(total, device=dev) - start[box]
    ys, zs = dims[box, 1], dims[box, 2]
    coors = torch.stack([local // (ys * zs), (local // zs) % ys, local % zs], 1)
    centers = coors.to(torch.float) * voxel_size + (-size / 2)[box] + voxel_size / 2
    return centers, box, k
