#!/usr/bin/env python3
"""Training entry point with the CLI of the reference's tools/train.py (:27-99):
    python tools/train.py <config> [--work-dir DIR] [--resume-from CKPT] [--no-validate]
                          [--cfg-options k=v ...] [--launcher {none,pytorch}] [--seed N]
One process per GPU (torchrun-style env), RCCL gradient all-reduce, AdamW + cosine schedule of
configs/_base_/schedules/cosine_2x.py, checkpoints with the reference's parameter names.

Data: by default Waymo-shaped synthetic tracklets generated in memory; with --data-root DIR the tracklet dataset
and the ococcnet.py train pipeline (objectcentricocccompletion_amd/dataset.py, pipelines.py) read files in the
reference's on-disk formats (tools/make_synthetic_dataset.py writes a small set of them; a real
data/waymo tree prepared by the reference's converters has the same layout), sharded over the ranks."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser(description='Train OcOccNet (MI355X)')
    ap.add_argument('config')
    ap.add_argument('--work-dir', default='work_dirs/ococcnet')
    ap.add_argument('--resume-from')
    ap.add_argument('--no-validate', action='store_true')
    ap.add_argument('--cfg-options', nargs='+', default=[])
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--local_rank', type=int, default=0)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--synthetic', action='store_true', default=True)
    ap.add_argument('--data-root', default=None, help='tree with tracklet_data/*.pkl, poses.pkl, occ_gt/ (see the docstring)')
    ap.add_argument('--proposals', default='tracklet_data/synth_training.pkl')
    ap.add_argument('--candidates', default='tracklet_data/synth_training_gt_candidates.pkl')
    ap.add_argument('--occ-root', default='occ_gt')
    return ap.parse_args()


def main():
    args = parse_args()
    from objectcentricocccompletion_amd import config, heads, point_pool, roi_head  # noqa: F401
    from objectcentricocccompletion_amd.dist import GradBuckets, broadcast_parameters, init_dist
    from objectcentricocccompletion_amd.registry import DETECTORS
    from objectcentricocccompletion_amd.synthetic import synthetic_training_batch
    cfg = config.fromfile(args.config)
    opts = {}
    for kv in args.cfg_options:
        k, v = kv.split('=', 1)
        try:
            v = eval(v, {}, {})
        except Exception:
            pass
        opts[k] = v
    config.merge_from_dict(cfg, opts)
    rank, world, local_rank = init_dist() if args.launcher == 'pytorch' else (0, 1, 0)
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    torch.manual_seed(args.seed)
    model = DETECTORS.build(cfg['model']).to(dev)
    start = 0
    if args.resume_from:
        ck = torch.load(args.resume_from, map_location=dev)
        model.load_state_dict(ck['state_dict'])
        start = ck.get('meta', {}).get('iter', 0)
    broadcast_parameters(model)
    ocfg = dict(cfg.get('optimizer', dict(type='AdamW', lr=1e-6, weight_decay=0.01)))
    ocfg.pop('type', None)
    ocfg.pop('paramwise_cfg', None)
    opt = torch.optim.AdamW(model.parameters(), fused=True, **ocfg)
    clip = (cfg.get('optimizer_config') or {}).get('grad_clip', dict(max_norm=10, norm_type=2))
    buckets = GradBuckets(model.parameters())
    samples = cfg.get('data', {}).get('samples_per_gpu', 4)
    ds = None
    if args.data_root:
        import numpy as np
        from objectcentricocccompletion_amd import dataset  # noqa: F401 (registers)
        from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_train_pipeline
        from objectcentricocccompletion_amd.pipelines import collate_tracklets
        from objectcentricocccompletion_amd.registry import DATASETS
        j = lambda p: os.path.join(args.data_root, p)
        ds = DATASETS.build(dict(type='WaymoTrackletDatasetWithOcc', data_root=args.data_root, ann_file=j(args.candidates),
                                 tracklet_proposals_file=j(args.proposals), occ_anno_root=j(args.occ_root),
                                 pose_file=j('poses.pkl'), pipeline=ococcnet_train_pipeline(), classes=['Car'],
                                 min_tracklet_points=100, min_tracklet_length=32))
        np.random.seed(args.seed + rank)
        order = np.random.permutation(len(ds))
        if rank == 0:
            print(f'{len(ds)} tracklets under {args.data_root}', flush=True)
    model.train()
    for it in range(start, start + args.iters):
        if ds is not None:  # tracklets sharded over the ranks: rank r takes every world-th sample
            idx = [int(order[((it * world + rank) * samples + b) % len(ds)]) for b in range(samples)]
            batch = collate_tracklets([ds[i] for i in idx], dev)
        else:
            batch = synthetic_training_batch(samples, 32, seed=args.seed + it * world + rank, device=dev)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        total = sum(v.mean() for k, v in losses.items() if k.startswith('loss'))
        total.backward()
        buckets.all_reduce()
        if clip:
            torch.nn.utils.clip_grad_norm_(model.parameters(), **clip)
        opt.step()
        if rank == 0:
            torch.cuda.synchronize()
            print(f'iter {it + 1}: loss {float(total):.4f} cls {float(losses["loss_rcnn_cls"]):.4f} '
                  f'bbox {float(losses["loss_rcnn_bbox"]):.4f} occ {float(losses["loss_rcnn_occ"].mean()):.4f} '
                  f'({(time.perf_counter() - t0) * 1e3:.1f} ms)', flush=True)
    if rank == 0:
        os.makedirs(args.work_dir, exist_ok=True)
        torch.save(dict(state_dict=model.state_dict(), meta=dict(iter=start + args.iters)),
                   os.path.join(args.work_dir, 'latest.pth'))


if __name__ == '__main__':
    main()
