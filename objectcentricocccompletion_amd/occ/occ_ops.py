"""Host mirror of mmdet3d/ops/occ/occ_ops.py: voxel-centre helpers of the implicit
occupancy grid (generate_dense_voxel_centers :5-50, quantize_points :53-93,
jitter_voxel_center :96-100).  Elementwise index math on small tensors."""
import torch


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
    sizes = sizes * sizes.new_tensor(scale_wlh).view(1, 3) + sizes.new_tensor(offset_wlh).view(1, 3)
    min_bound = (-sizes / 2)[rois_points_idx.long()]
    voxel_coors = torch.floor((points - min_bound) / voxel_size).to(torch.long)
    if to_center:
        return voxel_coors.to(torch.float) * voxel_size + min_bound + voxel_size / 2
    return voxel_coors


def jitter_voxel_center(voxel_size, voxel_centers):
    return voxel_centers + torch.rand_like(voxel_centers) * voxel_size - voxel_size / 2
