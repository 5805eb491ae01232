#!/usr/bin/env python3
"""GT-occupancy annotation of Waymo tracklets -- command line of the reference's tools/occ/occ_annotate.py:200-225,
723-740 on objectcentricocccompletion_amd.occ.annotate.OccAnnotator (one process per GPU: --rank / --world shard the
segments; --workers / --ngpus of the reference are accepted and ignored)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--bin-path', type=str, default='./data/waymo/waymo_format/train_gt.bin')
    p.add_argument('--split', type=str, default='training')
    p.add_argument('--workers', type=int, default=1)
    p.add_argument('--ngpus', type=int, default=8)
    p.add_argument('--chunksize', type=int, default=1)
    p.add_argument('--voxel-size', type=float, default=0.2)
    p.add_argument('--type', type=str, default='vehicle')
    p.add_argument('--data-root', type=str, default='./data/waymo/')
    p.add_argument('--out-dir', type=str, default='./work_dirs/occ_annotate/waymo_occ_gt')
    p.add_argument('--debug', action='store_true', default=False)
    p.add_argument('--cpu-voxelization', action='store_true', default=False)
    p.add_argument('--save-mean-var', action='store_true', default=False)
    p.add_argument('--overwrite', action='store_true', default=False)
    p.add_argument('--rank', type=int, default=int(os.environ.get('RANK', 0)))
    p.add_argument('--world', type=int, default=int(os.environ.get('WORLD_SIZE', 1)))
    a = p.parse_args(argv)
    import torch
    from objectcentricocccompletion_amd.occ.annotate import OccAnnotator
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    ann = OccAnnotator(a.data_root, a.out_dir, a.split, a.voxel_size, a.bin_path, a.type, a.workers, a.debug, a.cpu_voxelization,
                       a.overwrite, a.save_mean_var, a.ngpus, a.rank, a.world)
    n = ann.annotate_segment(a.chunksize)
    print(f'rank {a.rank}: {n} tracklets annotated')


if __name__ == '__main__':
    main()
