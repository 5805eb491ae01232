"""Kernel choice from a DEVICE-measured rulebook density (spconv.ops.DensityTracker) and one captured
graph serving batches of very different density."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from test_gpu_spconv import _voxels

pytestmark = pytest.mark.gpu


def test_density_is_measured_on_the_device_without_a_hint(dev):
    """no host number.  The tracker observes every rulebook build (asynchronous copy of the pair count
    behind an event) and the next builds carry the estimate; sparse and dense tables select different kernels."""
    from objectcentricocccompletion_amd.spconv import ops
    assert ops.DEFAULT_PAIRS_PER_ROW is None and ops.AUTO_DENSITY
    ops.density.reset()
    rng = np.random.default_rng(3)
    try:
        for dens, lo, hi, tile in ((0.03, 1.0, 2.5, True), (0.8, 12.0, 27.0, False)):
            ops.density.reset()
            idx = _voxels(rng, 4, (20, 20, 20), dens, False)
            t = torch.from_numpy(idx).to(dev)
            _, pairs, num = ops.get_indice_pairs(t, 4, [20, 20, 20], 3, subm=True)
            assert getattr(pairs._ococc, 'pairs_per_row', None) is None     # first build: nothing harvested yet
            torch.cuda.synchronize()
            est = ops.density.poll()
            true = float(num.sum().item()) / len(idx)
            assert est is not None and abs(est - true) < 1e-6 and lo < est < hi
            _, pairs2, _ = ops.get_indice_pairs(t, 4, [20, 20, 20], 3, subm=True)
            assert abs(pairs2._ococc.pairs_per_row - true) < 1e-6
            assert ops._use_tile_kernel(pairs2._ococc, 128, 64) == tile
            assert ops.density_regime(128, 64) == ('tile' if tile else 'stationary')
    finally:
        ops.density.reset()


def test_one_captured_graph_serves_sparse_and_dense_batches(dev):
    """Two batches of very different density through ONE captured graph (rulebook + 64 -> 32 convolution on a fixed
    number of rows): whatever kernel the capture-time density selected, both results equal the oracle."""
    from objectcentricocccompletion_amd.spconv import ops
    rng = np.random.default_rng(11)
    shape, B, n = (16, 16, 16), 2, 1500

    def batch(clustered):
        if clustered:   # a solid block of cells: ~20 neighbours per voxel
            cells = [(b, z, y, x) for b in range(B) for z in range(2, 11) for y in range(2, 11) for x in range(2, 12)]
            idx = np.array(cells[:n], dtype=np.int32)
        else:           # scattered: < 1 neighbour per voxel
            flat = np.sort(rng.choice(B * 16 ** 3, n, replace=False))
            idx = np.stack([flat // 4096, (flat // 256) % 16, (flat // 16) % 16, flat % 16], 1).astype(np.int32)
        return idx

    w = O.bf16_round(rng.standard_normal((3, 3, 3, 64, 32)).astype(np.float32) * 0.2)
    wt = torch.from_numpy(w).to(dev)
    idx_dev = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    x_dev = torch.zeros((n, 64), dtype=torch.float32, device=dev)
    for capture_on in (False, True):       # capture under the sparse density, then under the dense one
        ops.density.reset()
        first = batch(capture_on)
        idx_dev.copy_(torch.from_numpy(first))
        for _ in range(2):                  # eager builds: the tracker learns this density
            ops.get_indice_pairs(idx_dev, B, list(shape), 3, subm=True)
            torch.cuda.synchronize()
            ops.density.poll()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                _, pairs, num = ops.get_indice_pairs(idx_dev, B, list(shape), 3, subm=True)
                y_dev = ops.indice_conv(x_dev, wt, pairs, num, n, False, True)
        torch.cuda.current_stream().wait_stream(s)
        for clustered in (False, True, False):
            idx = batch(clustered)
            x = O.bf16_round(rng.standard_normal((n, 64)).astype(np.float32))
            idx_dev.copy_(torch.from_numpy(idx))
            x_dev.copy_(torch.from_numpy(x))
            g.replay()
            torch.cuda.synchronize()
            ep, en = O.subm_rulebook(idx, B, shape)
            ey = O.indice_conv(x, w, ep, en, n, subm=True)
            assert np.allclose(y_dev.cpu().numpy(), ey, rtol=1e-4, atol=2e-4), (capture_on, clustered)
    ops.density.reset()
