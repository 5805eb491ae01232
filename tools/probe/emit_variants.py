"""cost of the row-order bookkeeping inside grid_emit_kernel: time of object_grid_geometry with the library built as
shipped and with parts of the bookkeeping compiled out (OCOCC_LIB_PATH selects the build; wrong orders, timing only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from objectcentricocccompletion_amd.occ_encoder import synthetic_object_grids
from objectcentricocccompletion_amd.spconv import ops
from objectcentricocccompletion_amd.voxel import object_grid_geometry
dev = torch.device('cuda:0')
SLICES = int(sys.argv[1]) if len(sys.argv) > 1 else None
xyz, feats, bidx = synthetic_object_grids(64, 2000, seed=0, device=dev)
for label, ppr in (('no order', 9.0), ('with order', 1.8)):
    ops.DEFAULT_PAIRS_PER_ROW = ppr
    fn = lambda: object_grid_geometry(xyz, bidx, feats, [0.2] * 3, [-4, -4, -4, 4, 4, 4], [40, 40, 40], 64, out_dtype=torch.bfloat16, slices=SLICES)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print(f'{os.environ.get("OCOCC_LIB_PATH", "shipped")[-20:]:22s} {label:12s} {a.elapsed_time(b) * 10:7.1f} us per geometry (mark + bases + emit + place)')
