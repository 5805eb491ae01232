"""filter_pos_assigned_but_empty_rois without the per-sample loop: same rows as the loop of
fsd_bbox_head.py:442-455 (restated below) on sorted and unsorted sample indices."""
import pytest
import torch

from objectcentricocccompletion_amd.heads import SparseHeadMixin


def _loop(pos_data, pos_batch_idx, filtered_pos_mask, roi_batch_idx):
    out = []
    for b in range(int(roi_batch_idx.max()) + 1):
        keep = torch.nonzero(filtered_pos_mask[roi_batch_idx == b]).reshape(-1)
        out.append(pos_data[pos_batch_idx == b][keep])
    return torch.cat(out, 0)


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('sorted_idx', [True, False])
def test_rows_equal_reference_loop(seed, sorted_idx):
    g = torch.Generator().manual_seed(seed)
    B, n = 5, 200
    rb = torch.randint(0, B, (n,), generator=g)
    if seed == 0:
        rb[rb == 2] = 3  # a sample without RoIs
    if sorted_idx:
        rb = rb.sort().values
    # pos_data holds one row per RoI of each sample (the in-sample positions index it), in its own row order
    perm = torch.randperm(n, generator=g) if not sorted_idx else torch.arange(n)
    pb = rb[perm]
    data = torch.randn(n, 3, 2, generator=g)
    mask = torch.rand(n, generator=g) > 0.6
    head = SparseHeadMixin()
    want = _loop(data, pb, mask, rb.int())
    got = head.filter_pos_assigned_but_empty_rois(data, pb, mask, rb.int())
    assert torch.equal(got, want)
    # second tensor through the cached rows
    rbi = rb.int()
    a = head.filter_pos_assigned_but_empty_rois(data, pb, mask, rbi)
    b = head.filter_pos_assigned_but_empty_rois(data[:, 0], pb, mask, rbi)
    assert torch.equal(a, want) and torch.equal(b, want[:, 0])
    mask2 = ~mask
    assert torch.equal(head.filter_pos_assigned_but_empty_rois(data, pb, mask2, rbi), _loop(data, pb, mask2, rbi))


def test_in_place_change_of_the_mask_invalidates_the_cache():
    head = SparseHeadMixin()
    rb = torch.tensor([0, 0, 1, 1, 1]).int()
    pb = rb.long()
    data = torch.arange(5.)
    mask = torch.tensor([True, False, True, True, False])
    assert head.filter_pos_assigned_but_empty_rois(data, pb, mask, rb).tolist() == [0., 2., 3.]
    mask[1] = True
    assert head.filter_pos_assigned_but_empty_rois(data, pb, mask, rb).tolist() == [0., 1., 2., 3.]
    none = torch.zeros(5, dtype=torch.bool)
    assert head.filter_pos_assigned_but_empty_rois(data, pb, none, rb).numel() == 0
